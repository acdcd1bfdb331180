#!/usr/bin/env python3
"""Rehearsal on a 1-GPU box: N ranks (gloo, all on cuda:0) render their tile shards of one frame and assemble it on rank 0
with both exchange forms of gltf_renderer_amd/sharding.py; rank 0 also renders the whole frame alone.  All three must be
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
    from gltf_renderer_amd.sharding import TileExchange, reduce_frame
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
    TileExchange(s.width, s.height, world, "cpu").gather_frame(gathered, rank)
    tx, ty = (s.width + 15) // 16, (s.height + 15) // 16
    y, x = np.mgrid[0:s.height, 0:s.width]
    own = torch.from_numpy((((y // 16) * tx + (x // 16)) % world) == rank)
    summed = torch.where(own[..., None], host, torch.zeros_like(host))
    reduce_frame(summed, world)
    if rank == 0:
        full = r.create_output(s.width, s.height)
        r.trace(st, s.execute_params(frame=5, env_handle=h["env"]), full)
        torch.cuda.synchronize()
        f = full.cpu()
        ok_g, ok_r = bool(torch.equal(gathered, f)), bool(torch.equal(summed, f))
        print("world %d: gather == full frame: %s, reduce == full frame: %s" % (world, ok_g, ok_r))
        if not (ok_g and ok_r):
            sys.exit(1)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
