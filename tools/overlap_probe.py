#!/usr/bin/env python3
"""Run ON THE GPU BOX: do two independent sample batches on two HIP streams overlap usefully (the shade stage of one under the
traversal stage of the other)?  One context at 2K samples per launch against two contexts at K each, enqueued back to back.
usage: python tools/overlap_probe.py [sponza|grid] [K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gltf_renderer_amd import abi, scenes
from gltf_renderer_amd.renderer import Renderer

which = sys.argv[1] if len(sys.argv) > 1 else "sponza"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
s = {"sponza": scenes.sponza_class, "grid": scenes.material_grid}[which]()

def setup(stream, spp):
    r = Renderer(stream=stream.cuda_stream if stream is not None else None)
    h = s.upload(r); r.build_accel(); r.set_samples_per_trace(spp)
    return r, h, r.create_output(s.width, s.height)

def run(ctxs, frames):
    st = abi.PtSettings.from_buffer_copy(bytes(s.settings)); st.reset = 0
    for f in range(frames):
        for (r, h, out) in ctxs:
            r.trace(st, s.execute_params(f * 8, env_handle=h["env"]), out)

def measure(ctxs, label, frames=6):
    run(ctxs, 3); torch.cuda.synchronize()
    for c in ctxs: c[0].reset_stats()
    import time
    t0 = time.perf_counter(); run(ctxs, frames); torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
    rays = sum(c[0].stats().rays for c in ctxs)
    print("%-34s %.3f ms per round, %.0f Mrays/s" % (label, ms / frames, rays / ms / 1e3), flush=True)

one = [setup(None, 2 * K)]
measure(one, "one context, %d spp per launch" % (2 * K))
one[0][0].set_samples_per_trace(K)
measure(one, "one context, %d spp per launch" % K)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
two = [setup(s1, K), setup(s2, K)]
measure(two, "two contexts / streams, %d spp each" % K)
three = two + [setup(torch.cuda.Stream(), K)]
measure(three, "three contexts / streams, %d spp each" % K)
