#!/usr/bin/env python3
"""Run ON THE GPU BOX: does a single-sample frame finish sooner as P tile shards on P HIP streams (one context each, disjoint tiles of
the same output image) than as one launch sequence?  The stage tails of one shard can be filled by the other's stages.
usage: python tools/split_probe.py [sponza|grid] [spp]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gltf_renderer_amd import abi, scenes
from gltf_renderer_amd.renderer import Renderer

which = sys.argv[1] if len(sys.argv) > 1 else "sponza"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1
s = {"sponza": scenes.sponza_class, "grid": scenes.material_grid}[which]()

def setup(stream):
    r = Renderer(stream=stream.cuda_stream if stream is not None else None)
    h = s.upload(r); r.build_accel(); r.set_samples_per_trace(K)
    return r, h

def run(ctxs, out, frames):
    st = abi.PtSettings.from_buffer_copy(bytes(s.settings)); st.reset = 0
    P = len(ctxs)
    for f in range(frames):
        for k, (r, h) in enumerate(ctxs):
            r.trace(st, s.execute_params(f * K, tile_rank=k, tile_rank_count=P, env_handle=h["env"]), out)

def measure(ctxs, label, frames=20):
    out = ctxs[0][0].create_output(s.width, s.height)
    run(ctxs, out, 3); torch.cuda.synchronize()
    for c in ctxs: c[0].reset_stats()
    t0 = time.perf_counter(); run(ctxs, out, frames); torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
    rays = sum(c[0].stats().rays for c in ctxs)
    print("%-40s %.3f ms per frame, %.0f Mrays/s" % (label, ms / frames, rays / ms / 1e3), flush=True)

measure([setup(None)], "one context, whole frame, %d spp" % K)
for P in (2, 3, 4):
    measure([setup(torch.cuda.Stream()) for _ in range(P)], "%d tile shards on %d streams, %d spp" % (P, P, K))
