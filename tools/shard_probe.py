#!/usr/bin/env python3
"""What one rank of an N-GPU run does, measured on one MI355X: render rank 0's tile shard (1/N of the tiles) of the
Sponza-class 1080p frame, (a) one sample per launch (strong scaling: 1/N of the work) and (b) N samples in one sample
batch (weak scaling: the work of a 1-GPU step).  Shows why the bench batches: a 1/8 shard of one sample cannot fill
256 CUs."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gltf_renderer_amd import abi, scenes
from gltf_renderer_amd.renderer import Renderer

s = scenes.sponza_class()
r = Renderer(0)
h = s.upload(r)
r.build_accel()
out = r.create_output(s.width, s.height)
st = abi.PtSettings.from_buffer_copy(bytes(s.settings))
st.reset = 1
K = 12
for n in (1, 2, 4, 8):
    for spp in sorted({1, n, min(8 * n, 64)}):          # 8 x N is what bench.py gives each rank of an N-GPU run
        r.set_samples_per_trace(spp)
        for f in range(2):
            r.trace(st, s.execute_params(frame=f * spp, tile_rank=0, tile_rank_count=n, env_handle=h["env"]), out)
        torch.cuda.synchronize()
        r.reset_stats()
        t0 = time.perf_counter()
        for f in range(K):
            r.trace(st, s.execute_params(frame=(10 + f) * spp, tile_rank=0, tile_rank_count=n, env_handle=h["env"]), out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        c = r.stats()
        print("rank 0 of %d, %d spp/launch: %7.3f ms/launch  %6.2f M rays/launch  %6.0f Mrays/s per GPU  (x%d ranks = %6.0f)"
              % (n, spp, dt * 1e3, c.rays / K / 1e6, c.rays / K / dt / 1e6, n, n * c.rays / K / dt / 1e6))
