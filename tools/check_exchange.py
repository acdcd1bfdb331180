#!/usr/bin/env python3
"""Rehearsal on a 1-GPU box: N ranks (gloo, all on cuda:0) render their tile shards of one frame and assemble it on rank 0
with both exchange forms of the torch.distributed test double (tests/sharding_double.py) -- and moves the tiles through
libmipt.so's own pack / unpack kernels (pt_tiles_pack / pt_tiles_unpack, the device half of pt_exchange_frame) --; rank 0 also renders the whole frame alone.  All three must be
bit-identical.  Launch:  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/check_exchange.py"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    from gltf_renderer_amd import abi, scenes
    from gltf_renderer_amd.renderer import Renderer
    from tests.sharding_double import TileExchange
    s = scenes.test_scene(200, 64)
    s.width, s.height = 200, 136                       # partial edge tiles in both directions
    r = Renderer(0)
    h = s.upload(r)
    st = abi.PtSettings.from_buffer_copy(bytes(s.settings))
    st.reset = 1
    mine = r.create_output(s.width, s.height)
    mine.fill_(777.0)                                  # garbage outside my tiles: the gather must not read it
    r.trace(st, s.execute_params(frame=5, tile_rank=rank, tile_rank_count=world, env_handle=h["env"]), mine)
    torch.cuda.synchronize()
    host = mine.cpu()
    gathered = host.clone()
    xch = TileExchange(s.width, s.height, world, "cpu")
    xch.gather_frame(gathered, rank)
    summed = xch.reduce_frame(host, rank)
    # the library's own packing: my tiles packed on the GPU, gathered over gloo, unpacked on rank 0's GPU
    packed = r.tiles_pack(mine, rank, world)
    torch.cuda.synchronize()
    sizes = [r.tiles_packed_bytes(s.width, s.height, k, world) // 16 for k in range(world)]
    pad = torch.zeros((max(sizes), 4), dtype=torch.float32)
    pad[: sizes[rank]] = packed.cpu()
    parts = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, parts, dst=0)
    if rank == 0:
        frame = torch.full_like(mine, float("nan"))
        for k in range(world):
            r.tiles_unpack(parts[k][: sizes[k]].cuda(), frame, k, world)
        torch.cuda.synchronize()
    if rank == 0:
        full = r.create_output(s.width, s.height)
        r.trace(st, s.execute_params(frame=5, env_handle=h["env"]), full)
        torch.cuda.synchronize()
        f = full.cpu()
        ok_g, ok_r, ok_p = bool(torch.equal(gathered, f)), bool(torch.equal(summed, f)), bool(torch.equal(frame.cpu(), f))
        print("world %d: gather == full frame: %s, reduce == full frame: %s, pt_tiles_pack/unpack == full frame: %s" % (world, ok_g, ok_r, ok_p))
        if not (ok_g and ok_r and ok_p):
            sys.exit(1)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
