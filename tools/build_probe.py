"""Warm rebuild time of the acceleration structure (pt_build_accel on a scene whose scratch is already allocated).
usage: python tools/build_probe.py [repeats] [sponza|figure]      (run under `rocprofv3 --kernel-trace --stats` for the per-kernel split)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gltf_renderer_amd import scenes
from gltf_renderer_amd.renderer import Renderer

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
s = (scenes.skinned_figure if (len(sys.argv) > 2 and sys.argv[2] == 'figure') else scenes.sponza_class)(); r = Renderer(device=0); h = s.upload(r)
r.build_accel(); torch.cuda.synchronize()
wall, dev = [], []
for i in range(n):
    t = time.perf_counter(); r.build_accel(); torch.cuda.synchronize(); wall.append((time.perf_counter() - t) * 1e3)
    dev.append(r.stats().accel_ms)
print("triangles %d  wide nodes %d" % (s.triangles, r.stats().bvh_nodes))
print("rebuild wall ms:", " ".join("%.3f" % x for x in wall))
print("rebuild accel_ms:", " ".join("%.3f" % x for x in dev))
