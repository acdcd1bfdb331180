#!/usr/bin/env python3
"""Run ON THE GPU BOX: the builders side by side on a scene -- build time, node count, stack bound, nodes / triangles per ray,
Mrays/s at 1 and 8 samples per launch -- and the images must agree (same closest hits: bit-identical up to exact-t ties).
usage: python tools/builder_probe.py [sponza|helmet|grid|figure|test]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gltf_renderer_amd import abi, scenes
from gltf_renderer_amd.renderer import Renderer

which = sys.argv[1] if len(sys.argv) > 1 else "sponza"
s = {"sponza": scenes.sponza_class, "helmet": scenes.helmet_class, "grid": scenes.material_grid, "figure": scenes.skinned_figure,
     "test": lambda: scenes.test_scene(512, 256)}[which]()
imgs = {}
for name, b in (("lbvh", abi.BUILDER_LBVH), ("ploc", abi.BUILDER_PLOC), ("reins", abi.BUILDER_PLOC_REINSERT)):
    r = Renderer(); r.set_accel_builder(b)
    h = s.upload(r)
    r.build_accel(); r.request_rebuild(); r.build_accel(); torch.cuda.synchronize()
    q = r.stats()
    st = abi.PtSettings.from_buffer_copy(bytes(s.settings)); st.debug_output = abi.DEBUG_OUTPUT_TEXCOORD_0; st.flags &= ~abi.FLAG_ACCUMULATE; st.use_frame_as_seed = 0; st.seed = 3
    out = r.create_output(s.width, s.height)
    r.trace(st, s.execute_params(0, env_handle=h["env"]), out)
    imgs[name] = r.readback(out)
    line = "%s %-5s build %.3f ms  wide nodes %d  stack need %d" % (s.name, name, q.accel_ms, q.bvh_nodes, q.bvh_stack_need)
    for spp in (1, 8):
        r.set_samples_per_trace(spp)
        st = abi.PtSettings.from_buffer_copy(bytes(s.settings)); st.reset = 1
        out = r.create_output(s.width, s.height)
        for f in range(3):
            r.trace(st, s.execute_params(f * spp, env_handle=h["env"]), out); st.reset = 0
        torch.cuda.synchronize(); r.reset_stats()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for f in range(3, 9):
            r.trace(st, s.execute_params(f * spp, env_handle=h["env"]), out)
        e.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(e) / 6
        line += "  | %d spp: %.3f ms/launch %.0f Mrays/s" % (spp, ms, r.stats().rays / 6 / ms / 1e3)
    r.enable_counters(True); r.reset_stats(); r.set_samples_per_trace(1)
    r.trace(st, s.execute_params(99, env_handle=h["env"]), out)
    c = r.stats()
    line += "  | nodes/ray %.2f tris/ray %.2f" % (c.nodes_visited / c.rays, c.tris_tested / c.rays)
    print(line)
    r.close()
for other in ("ploc", "reins"):
    d = np.abs(imgs["lbvh"] - imgs[other]).max(axis=2)
    print("texcoord debug image: pixels that differ between lbvh and %s: %.5f" % (other, float((d > 0).mean())))
