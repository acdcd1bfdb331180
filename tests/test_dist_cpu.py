"""Multi-rank tile sharding on CPU: world_size 2 over gloo.  Each rank renders its tiles (tile t belongs to rank
t % N) into a zeroed full-size image and one reduce(sum) to rank 0 assembles the frame.
The compute on the CPU is the oracle (tests may use it); the sharding / reduce logic is what is under test
and is the same code path bench.py runs over RCCL."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gltf_renderer_amd import abi, scenes
    from tests.sharding_double import my_tile_count
    from oracle import pyoracle
    s = scenes.test_scene(40, 16)
    st = abi.PtSettings.from_buffer_copy(bytes(s.settings)); st.flags &= ~abi.FLAG_ACCUMULATE
    o = pyoracle.Oracle(); h = s.upload(o)
    img = np.zeros((s.height, s.width, 4), np.float32)
    o.trace(st, s.execute_params(frame=3, env_handle=h["env"], tile_rank=rank, tile_rank_count=world), img, nthreads=1)
    touched = int((img[..., 3] != 0).sum())
    t = torch.from_numpy(img)
    # the point-to-point form of the exchange on a copy whose foreign pixels hold garbage (it must not read them) ...
    from tests.sharding_double import TileExchange
    g = torch.from_numpy(np.where(img[..., 3:4] != 0, img, np.float32(123.0)).astype(np.float32))
    xch = TileExchange(s.width, s.height, world, "cpu")
    xch.gather_frame(g, rank)
    if rank == 0:
        np.save(os.path.join(tmp, "gathered.npy"), g.numpy())
    # ... and the reduce(sum) form, on a zero-masked copy: the rank's own image is not touched
    g2 = torch.from_numpy(np.where(img[..., 3:4] != 0, img, np.float32(np.nan)).astype(np.float32))
    before = g2.clone()
    red = xch.reduce_frame(g2, rank)
    assert torch.equal(torch.nan_to_num(g2, nan=7.0), torch.nan_to_num(before, nan=7.0))
    counts = torch.tensor([touched, my_tile_count(s.width, s.height, rank, world)])
    dist.all_reduce(counts)
    if rank == 0:
        np.save(os.path.join(tmp, "sharded.npy"), red.numpy())
        np.save(os.path.join(tmp, "counts.npy"), counts.numpy())
    # ---- accumulation across frames composes with both exchanges (each rank's accumulation image stays private): three
    # accumulated frames, one exchange after each, every assembled frame must equal the 1-rank running mean
    st2 = abi.PtSettings.from_buffer_copy(bytes(s.settings)); st2.reset = 1
    acc = np.zeros((s.height, s.width, 4), np.float32)
    for f in range(3):
        o.trace(st2, s.execute_params(frame=10 + f, env_handle=h["env"], tile_rank=rank, tile_rank_count=world), acc, nthreads=1)
        st2.reset = 0
        a_t = torch.from_numpy(acc)
        frame_g = xch.gather_frame(a_t, rank, out=torch.empty_like(a_t) if rank == 0 else None)
        frame_r = xch.reduce_frame(a_t, rank)
        if rank == 0:
            np.save(os.path.join(tmp, "acc_gather_%d.npy" % f), frame_g.numpy())
            np.save(os.path.join(tmp, "acc_reduce_%d.npy" % f), frame_r.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_tile_sharding_equals_single_rank(tmp_path, oracle_lib):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    from gltf_renderer_amd import abi, scenes
    s = scenes.test_scene(40, 16)
    st = abi.PtSettings.from_buffer_copy(bytes(s.settings)); st.flags &= ~abi.FLAG_ACCUMULATE
    o = oracle_lib.Oracle(); h = s.upload(o)
    full = np.zeros((s.height, s.width, 4), np.float32)
    o.trace(st, s.execute_params(frame=3, env_handle=h["env"]), full, nthreads=1)
    sharded = np.load(tmp_path / "sharded.npy")
    assert np.array_equal(sharded, full)                    # bit for bit
    assert np.array_equal(np.load(tmp_path / "gathered.npy"), full)
    counts = np.load(tmp_path / "counts.npy")
    assert counts[0] == 40 * 40 and counts[1] == 9          # every pixel rendered exactly once; 3x3 tiles
    # accumulate + exchange over three frames == the 1-rank running mean after each frame (an in-place reduce of the
    # accumulation target double-counted the other ranks' tiles from the second frame on)
    st2 = abi.PtSettings.from_buffer_copy(bytes(s.settings)); st2.reset = 1
    acc = np.zeros((s.height, s.width, 4), np.float32)
    for f in range(3):
        o.trace(st2, s.execute_params(frame=10 + f, env_handle=h["env"]), acc, nthreads=1)
        st2.reset = 0
        assert np.array_equal(np.load(tmp_path / ("acc_gather_%d.npy" % f)), acc), f
        assert np.array_equal(np.load(tmp_path / ("acc_reduce_%d.npy" % f)), acc), f


def test_tile_partition_is_exact():
    from tests.sharding_double import my_tile_count, tile_owner
    for (w, h) in ((1920, 1080), (3840, 2160), (50, 17), (16, 16), (1, 1)):
        tiles = ((w + 15) // 16) * ((h + 15) // 16)
        for n in (1, 2, 3, 4, 8):
            assert sum(my_tile_count(w, h, r, n) for r in range(n)) == tiles
            owners = [tile_owner(t, n) for t in range(tiles)]
            assert all(0 <= o < n for o in owners)
            for r in range(n):
                assert owners.count(r) == my_tile_count(w, h, r, n)


def test_bench_refuses_a_rank_count_it_cannot_honour():
    """`python bench.py --gpus N` started plainly launches its own ranks -- and exits non-zero, before any GPU call, when the node has
    fewer than N devices; under a launcher whose WORLD_SIZE differs from --gpus it refuses too: `n_gpus` of the line is always the
    number of ranks that rendered (it used to render on one GPU and print n_gpus 1)."""
    import subprocess
    if torch.cuda.device_count() >= 2:
        pytest.skip("two devices present: the launch would succeed")
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "device(s) visible" in out.stderr and not out.stdout.strip()
    out = subprocess.run([sys.executable, bench, "--gpus", "1"], env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "WORLD_SIZE 2" in out.stderr and not out.stdout.strip()
