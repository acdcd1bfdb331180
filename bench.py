#!/usr/bin/env python3
"""Headline benchmark: Mrays/s + ms/frame of the path-tracing hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched under torch.distributed.run, one rank per
GPU.  Workload = BASELINE.json configs[2], the config the metric "@1920x1080, 8-bounce" is quoted on: the Sponza-class stand-in,
1920x1080, 8 bounces + RR, punctual lights + env MIS (configs[1], [3], [4] are parity-test cases: tests/test_gpu_round2.py).

A STEP is one `pt_trace` of the whole frame carrying a SAMPLE BATCH of 8 samples per pixel (pt_set_samples_per_trace): one set
of kernel launches, bit-identical to 8 consecutive reference frames (tested).  The reference renders one sample per
PathtraceScene call because it presents every frame; an offline renderer accumulates hundreds, and a batch keeps 256 CUs full
through the thin late bounces.  The reference-semantics figure -- one 1-spp `pt_trace`, SURVEY 8(d) -- is reported beside it as
`config.ms_single_sample_launch` (median of >= 20 launches after 3 warm-ups) with `config.mrays_single_sample_launch`.

N > 1: the frame is sharded by 16x16 pixel tile (tile t -> rank t % N), every rank renders its tiles into ITS OWN accumulation
image, and ONE exchange per step assembles the frame on rank 0 -- `pt_exchange_frame` inside libmipt.so (RCCL on the launch
stream: every rank sends its own tiles point to point over its direct xGMI link, or `--exchange reduce`: ncclReduce of a
zero-masked copy).  The headline line is WEAK scaling, as the partitioned path calls for: every rank keeps the per-GPU work of the N = 1
step -- its 1/N of the tiles carries 8 x N samples per step (the frame is the same 1920x1080; N GPUs deliver N times the samples per
unit time), `"scaling": "weak"`.  The same run then times the STRONG variant (the 8-spp step split N ways: total work fixed, each rank
left with the work of one 1-spp frame) and reports it in `strong_scaling`; `--headline strong` swaps the two.  `python bench.py --gpus N` started plainly (no WORLD_SIZE) launches its own N ranks with torch.distributed.run as a
child process before anything touches a GPU, and exits non-zero if the node has fewer than N devices; the line's `n_gpus` is always the
number of ranks that rendered.  If RCCL cannot be brought up inside libmipt.so on every rank the run FAILS (no fallback transport).
`--backend gloo --single-device` rehearses the rank logic on one GPU through the torch.distributed test double
(tests/sharding_double.py) and says so in the line; RCCL itself needs one GPU per rank.  The N > 1 branches of the library's own
exchange_frame are tested on one GPU through its in-process loopback transport (tests/test_gpu_round3.py).

Timed region: inputs (scene, BVH, textures, env maps) resident in HBM; K steps bracketed by barrier + torch.cuda.synchronize on
both sides; time = max over ranks; value = rays traced by all ranks in the K steps / that time.  Rays are counted by the kernels
(every traversal started).  The accumulation is reset before it could reach max_accumulated_frames, so no timed step is a no-op.

Extra objects on the JSON line (rank 0, N = 1):
  roofline     - for the single DOMINANT KERNEL (largest share of a launch's GPU time; k_wf_traverse): achieved = its algorithmic
                 bytes per launch / its time per launch (HIP events after every stage launch on the launch stream, untimed replay
                 of the same frames; bytes from two counting replays, the second with max_bounces 0 to split off the primary
                 rays).  `peak` is a figure of MI355X_MICROARCH.md: the traversal kernels are cache-fed (PMC: the fabric moves a
                 fifth of their algorithmic bytes), so bound = "l2-gather" at the guide's 16.8 TB/s, with frac_of_hbm_peak
                 (8 TB/s) and frac_of_infinity_cache_gather (8.6 TB/s) beside it; the shade stage is priced against HBM.
                 `kernels` / `stages` hold the same for every kernel / kernel family, `own_probe_ceilings` the builder's own
                 measured ceilings (never the peak).  `traffic` = PMC-measured fabric bytes of that kernel per launch, collected IN THIS
                 RUN by two rocprofv3 --pmc child runs (FETCH_SIZE, WRITE_SIZE: separate passes) of a short version of the same command
                 (pmc_live); if the profiler is not available, from profiles/r03_pmc_per_kernel.json, stamped with its git head.
  cpu_baseline - the CPU oracle (kind "port": the reference has no CPU tracer and no CPU BVH build) on a bounded sample of the same
                 workload: all-core and 1-thread Mrays/s with the CPU model, plus `legs`: B2 CPU LBVH build vs the HIP build and
                 refit, B3 the restated per-frame host work (gs_animate + global transforms + gs_frame) in microseconds, B4 CPU
                 skinning vs the HIP kernels.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SPP_PER_STEP = 8            # samples per pixel per step of the headline line


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def median(v):
    v = sorted(v)
    return v[len(v) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="sponza", choices=["sponza", "helmet", "grid", "figure", "sponza_figure", "test"])
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-pmc-live", action="store_true", help="do not collect the dominant kernel's FETCH_SIZE / WRITE_SIZE in this run (two rocprofv3 --pmc child runs of a short "
                                                               "version of the same command); roofline.traffic then comes from the committed profile")
    ap.add_argument("--save-image", default="")
    ap.add_argument("--mode", default="wavefront", choices=["wavefront", "megakernel"])
    ap.add_argument("--stage-blocks", type=int, default=0)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo + --single-device rehearses N ranks on a 1-GPU box through the test double")
    ap.add_argument("--single-device", action="store_true", help="rehearsal only: every rank uses cuda:0 (RCCL needs one device per rank)")
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel per step (sample batch); default 8")
    ap.add_argument("--exchange", default="gather", choices=["gather", "reduce"], help="per-step assembly on rank 0: own-tile gather (default) or reduce of the zero-masked copy")
    ap.add_argument("--no-weak", action="store_true", help="N > 1: skip the second timed loop (the scaling mode that is not the headline)")
    ap.add_argument("--headline", default="weak", choices=["weak", "strong"], help="N > 1: which scaling mode the headline line reports (the other one rides in the same line)")
    ap.add_argument("--cull-null-shadow", action="store_true", help="experiment: pt_set_null_shadow_culling(1) -- same image, fewer shadow rays than the reference traces (off for the headline line)")
    ap.add_argument("--animate", action="store_true", help="dynamic path (--config figure / sponza_figure): skin -> BVH refit -> trace every step, accumulation reset each frame")
    ap.add_argument("--rebuild", action="store_true", help="with --animate: full LBVH rebuild every frame instead of the refit (the round-1 behaviour, for comparison)")
    args = ap.parse_args()

    # ---- `python bench.py --gpus N` with no launcher around it: start the N ranks ourselves, as a child process, BEFORE any GPU call
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import subprocess
        import torch
        have = torch.cuda.device_count()              # counting devices does not initialise the GPU
        if have < args.gpus and not args.single_device:
            raise SystemExit("bench.py: --gpus %d asked for, %d device(s) visible on this node" % (args.gpus, have))
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.call(cmd, env=env))

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:                            # the line's n_gpus is the number of ranks that render: never report another
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE %d (launch with --nproc-per-node %d, or start bench.py plainly and it launches its own ranks)"
                         % (args.gpus, world, args.gpus))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.single_device:
            local_rank = 0
        elif torch.cuda.device_count() < world:
            raise SystemExit("bench.py: %d ranks but %d device(s) visible" % (world, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")

    from gltf_renderer_amd import scenes, abi
    from gltf_renderer_amd.renderer import Renderer

    t_setup = time.time()
    if args.config == "sponza":
        s = scenes.sponza_class()
    elif args.config == "helmet":
        s = scenes.helmet_class()
    elif args.config == "grid":
        s = scenes.material_grid()
    elif args.config == "figure":
        s = scenes.skinned_figure()
    elif args.config == "sponza_figure":
        s = scenes.sponza_with_figure()
    else:
        s = scenes.test_scene(512, 256)
    if args.width and args.height:
        s.width, s.height = args.width, args.height

    r = Renderer(device=local_rank)
    r.set_kernel_mode(abi.MODE_MEGAKERNEL if args.mode == "megakernel" else abi.MODE_WAVEFRONT, args.stage_blocks)
    h = s.upload(r)
    binding = None
    if args.animate:
        if not s.skins:
            raise SystemExit("--animate needs a scene with a skinned mesh (--config figure or sponza_figure)")
        binding = scenes.SkinBinding(r, s, h, 0, use_mfma=1)      # every rank skins and refits itself
        binding.pose(0.0)
    r.build_accel()
    settings = s.settings
    base_spp = args.spp if args.spp > 0 else SPP_PER_STEP
    if args.animate:
        base_spp = 1                          # a playing animation resets accumulation every frame (Main.cpp:521-523)
    r.set_null_shadow_culling(args.cull_null_shadow)
    out = r.create_output(s.width, s.height)
    torch.cuda.synchronize()
    st_build = r.stats()
    accel_ms = st_build.accel_ms
    t_setup = time.time() - t_setup

    # ---- the one exchange per step: libmipt.so's own RCCL exchange; the torch.distributed double only for gloo rehearsals
    exchange_mode = args.exchange if world > 1 else "none"
    xch = None
    exchange_impl = "libmipt.so (RCCL)"
    if world > 1 and args.backend == "nccl":
        # The library's own exchange, or no run at all.  ncclCommInitRank is collective, so the ranks first AGREE that every one of them
        # can load RCCL (a probe that touches neither the GPU nor the network) before any of them enters it; every rank always takes part
        # in the broadcast of the id, whatever happened on rank 0.
        flag = torch.tensor([1 if Renderer.exchange_probe() else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            raise SystemExit("bench.py: librccl cannot be loaded inside libmipt.so on at least one rank (rank %d: %s): no exchange, no run"
                             % (rank, "ok" if Renderer.exchange_probe() else "failed"))
        ids = [None]
        if rank == 0:
            try:
                ids = [r.exchange_unique_id()]
            except Exception as e:                            # noqa: BLE001
                print("rank 0: pt_exchange_unique_id failed (%s)" % e, file=sys.stderr)
        dist.broadcast_object_list(ids, src=0)                # the host's transport for the 128-byte ncclUniqueId
        if ids[0] is None:
            raise SystemExit("bench.py: rank 0 could not make an ncclUniqueId")
        r.exchange_create(rank, world, ids[0])                # raises on failure: the launcher takes the job down
    elif world > 1:
        from tests.sharding_double import TileExchange
        xch = TileExchange(s.width, s.height, world, "cpu")
        exchange_impl = "torch.distributed test double over gloo"

    def exchange(img):
        if world == 1:
            return
        if xch is None:                     # product path: RCCL inside the library, on the launch stream, assembled in place on rank 0
            r.exchange_frame(img, None, mode=abi.EXCHANGE_GATHER if exchange_mode == "gather" else abi.EXCHANGE_REDUCE, dst=0)
        else:                               # rehearsal: gloo moves host tensors
            host = img.cpu()
            res = xch.gather_frame(host, rank) if exchange_mode == "gather" else xch.reduce_frame(host, rank)
            if rank == 0:
                img.copy_(res)

    def animate(frame):
        if binding is not None:             # Main.cpp:521-523: a playing animation resets accumulation every frame
            binding.pose((frame % 60) / 30.0)
            if args.rebuild:
                r.request_rebuild()
            r.build_accel()                 # refit (UpdateDynamicBlas), or the full rebuild with --rebuild
            settings.reset = 1

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    stat_dev = "cuda" if args.backend == "nccl" else "cpu"

    def timed_run(spp, first_frame):
        """warmup + K timed steps at `spp` samples per step.  Returns (elapsed max over ranks, rays of all ranks, per-step kernel ms)."""
        r.set_samples_per_trace(spp)
        state = {"acc": 0}
        settings.reset = 1

        def step(frame):
            animate(frame)
            if state["acc"] + spp > settings.max_accumulated_frames:     # never let a step become a no-op (Pathtracer.cpp:273)
                settings.reset = 1
            if settings.reset:
                state["acc"] = 0
            p = s.execute_params(frame=frame * spp, tile_rank=rank, tile_rank_count=world, env_handle=h["env"])
            r.trace(settings, p, out)
            settings.reset = 0
            state["acc"] += spp
            exchange(out)

        for f in range(args.warmup):
            step(first_frame + f)
        sync_all()
        r.reset_stats()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        sync_all()
        t0 = time.perf_counter()
        for k in range(args.steps):
            frame = first_frame + args.warmup + k
            animate(frame)
            if state["acc"] + spp > settings.max_accumulated_frames:
                settings.reset = 1
            if settings.reset:
                state["acc"] = 0
            p = s.execute_params(frame=frame * spp, tile_rank=rank, tile_rank_count=world, env_handle=h["env"])
            ev[k][0].record()
            r.trace(settings, p, out)
            ev[k][1].record()
            settings.reset = 0
            state["acc"] += spp
            exchange(out)
        sync_all()
        elapsed = time.perf_counter() - t0
        st = r.stats()
        el = torch.tensor([elapsed], dtype=torch.float64, device=stat_dev)
        ry = torch.tensor([float(st.rays)], dtype=torch.float64, device=stat_dev)
        if world > 1:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
            dist.all_reduce(ry, op=dist.ReduceOp.SUM)
        return float(el.item()), float(ry.item()), [a.elapsed_time(b) for a, b in ev]

    # ---- headline.  N = 1: the 8-spp step.  N > 1, weak scaling (default): per-GPU work as at N = 1 -- each rank's 1/N of the tiles carries
    # 8 x N samples per step; strong scaling (--headline strong): the 8-spp step split N ways
    weak_spp = min(base_spp * world, 64)              # (PT_MAX_SAMPLES_PER_TRACE = 64: N = 8 at 8 spp is exactly the limit)
    headline_weak = world > 1 and args.headline == "weak" and not args.animate
    spp = weak_spp if headline_weak else base_spp
    elapsed, rays_total, kernel_ms = timed_run(spp, 0)

    result = None
    if rank == 0:
        mrays = rays_total / elapsed / 1e6
        result = {
            "metric": "Mrays/sec + ms/frame @1920x1080, 8-bounce, 1/2/4/8 MI355X",
            "value": round(mrays, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1000.0, 4), "higher_is_better": True,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s %dx%d %dspp/step, max_bounces %d (limit %d), min_bounces %d, RR %.1f-%.1f, %d triangles / %d instances / %d textures / %d lights, env-map MIS, flags 0x%x"
                       % (s.name, s.width, s.height, spp, settings.max_bounces, s.bounce_limit, settings.min_bounces,
                          settings.min_russian_roulette_continue_prob, settings.max_russian_roulette_continue_prob, s.triangles,
                          len(s.instances), len(s.textures), len(s.lights), settings.flags),
                       "parallelism": ("tile-shard x%d + 1 exchange/step by %s (%s)" % (world, exchange_impl, "own-tile gather to rank 0, point to point" if exchange_mode == "gather" else "reduce of the zero-masked copy"))
                                      if world > 1 else "single GPU",
                       "null_shadow_culling": bool(args.cull_null_shadow),
                       "samples_per_step": spp, "ms_per_1spp_frame": round(elapsed / args.steps / spp * 1000.0, 4),
                       "rays_per_frame": round(rays_total / args.steps / spp, 1), "bvh_build_ms": round(accel_ms, 3),
                       "bvh_stack_need": int(st_build.bvh_stack_need), "scene_setup_s": round(t_setup, 1)},
        }
        # N = 1 is the first point of either series; it carries the mode of the series the N > 1 lines of the same command report
        result["scaling"] = "weak" if (args.headline == "weak" and not args.animate) else "strong"
        if world > 1:
            result["config"]["scaling_note"] = ("weak: per-GPU work fixed -- every rank renders its 1/%d of the tiles with %d samples per step (8 per GPU); "
                                                "strong_scaling = the 8-spp step split %d ways" % (world, spp, world)) if headline_weak else \
                                               ("strong: the %d-spp step split %d ways; weak_scaling = %d samples per step" % (spp, world, weak_spp))
            if args.backend != "nccl":
                result["config"]["parallelism"] += " [REHEARSAL over gloo on %s: not an RCCL measurement]" % ("one GPU" if args.single_device else "the host")

    # ---- N > 1: the other scaling mode of the same run
    if world > 1 and not args.no_weak and not args.animate:
        ospp = base_spp if headline_weak else weak_spp
        other = "strong" if headline_weak else "weak"
        w_elapsed, w_rays, _ = timed_run(ospp, 1000)
        if rank == 0:
            result[other + "_scaling"] = {"scaling": other, "value": round(w_rays / w_elapsed / 1e6, 3), "unit": "Mrays/s", "samples_per_step": ospp,
                                          "ms_per_step": round(w_elapsed / args.steps * 1000.0, 4), "ms_per_1spp_frame": round(w_elapsed / args.steps / ospp * 1000.0, 4)}
        r.set_samples_per_trace(spp)

    # ---- latency of a single-sample launch (one reference frame): median of >= 20 after 3 warm-ups
    if rank == 0 and world == 1 and binding is None and not args.no_roofline:     # (--no-roofline = profiling runs: timed launches only)
        r.set_samples_per_trace(1)
        lat, rays1 = [], 0
        settings.reset = 1
        for k in range(3 + max(20, args.steps)):
            if k == 3:
                r.reset_stats()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            r.trace(settings, s.execute_params(frame=100000 + k, env_handle=h["env"]), out)
            b.record()
            settings.reset = 0
            torch.cuda.synchronize()
            if k >= 3:
                lat.append(a.elapsed_time(b))
        rays1 = r.stats().rays / len(lat)
        result["config"]["ms_single_sample_launch"] = round(median(lat), 4)
        result["config"]["mrays_single_sample_launch"] = round(rays1 / (median(lat) * 1e-3) / 1e6, 1)
        result["config"]["single_sample_launches_timed"] = len(lat)
        r.set_samples_per_trace(spp)

    # ---- dynamic path: skin / accel (refit or rebuild) / trace split, measured on an untimed synchronised replay of a few frames
    if rank == 0 and binding is not None:
        parts = {"skin_ms": [], "accel_ms": [], "trace_ms": []}
        for k in range(max(args.steps, 8)):
            animate(args.warmup + k)
            r.trace(settings, s.execute_params(frame=(args.warmup + k) * spp, tile_rank=rank, tile_rank_count=world, env_handle=h["env"]), out)
            torch.cuda.synchronize()
            st2 = r.stats()
            parts["skin_ms"].append(st2.skin_ms); parts["accel_ms"].append(st2.accel_ms); parts["trace_ms"].append(st2.trace_ms)
        result["config"]["dynamic_ms"] = {k2: round(median(v), 4) for k2, v in parts.items()}
        st2 = r.stats()
        result["config"]["dynamic_ms"]["accel"] = "full rebuild every frame (--rebuild)" if args.rebuild else "refit (UpdateDynamicBlas); full builds so far: %d, refits: %d" % (st2.accel_builds, st2.accel_refits)
        result["config"]["dynamic_ms"]["first_build_ms"] = round(accel_ms, 4)
        result["config"]["parallelism"] += "; skin (MFMA joint blend) + %s + 1 spp trace per step" % ("full LBVH rebuild" if args.rebuild else "BVH refit")

    # ---- roofline (rank 0, N = 1): stage times from an untimed replay with an event after every stage launch (HIP events on the launch
    # stream, inside the library), then the same frames twice more with the node / triangle / hit / tap counters on -- once as they are
    # and once with max_bounces = 0, which isolates the primary rays' share (the counting kernels are slower, so they are never timed)
    if rank == 0 and world == 1 and not args.no_roofline and args.mode == "wavefront":
        settings2 = abi.PtSettings.from_buffer_copy(bytes(settings))
        out2 = r.create_output(s.width, s.height)
        n = float(args.steps)

        def replay(st, per_step=None):
            st.reset = 1
            for k in range(args.steps):
                animate(args.warmup + k)
                if binding is not None:
                    st.reset = 1
                r.trace(st, s.execute_params(frame=(args.warmup + k) * spp, env_handle=h["env"]), out2)
                st.reset = 0
                if per_step:
                    per_step()

        stage_ms = [0.0] * 5

        def add_stage_ms():
            q = r.stats()
            for i in range(5):
                stage_ms[i] += q.stage_ms[i] / n

        r.enable_stage_timing(True)
        replay(settings2, add_stage_ms)
        r.enable_stage_timing(False)
        r.enable_counters(True)
        torch.cuda.synchronize()
        r.reset_stats()
        replay(settings2)
        c = r.stats()
        settings0 = abi.PtSettings.from_buffer_copy(bytes(settings)); settings0.max_bounces = 0; settings0.min_bounces = 0
        r.reset_stats()
        replay(settings0)
        c0 = r.stats()                        # the same frames, primary rays only: k_wf_trace's node steps and triangle tests
        r.enable_counters(False)
        env_mis = bool(settings.flags & abi.FLAG_ENVIRONMENT_MIS) and bool(settings.flags & abi.FLAG_ENVIRONMENT_MAP)
        # bytes/ray = N_node*64 + N_tri*48 + [closest hits] S_hit + [misses] 64 + 32/R per pixel-sample; S_hit = 12 (indices)
        # + 72 (three vertices) + 176 (instance row) + 640 (material) + 16 per bilinear footprint + 160 (importance pyramid) with env MIS
        misses = c.rays_primary + c.rays_bounce - c.closest_hits
        primary_nodes = c0.nodes_visited - c0.nodes_visited_shadow
        primary_tris = c0.tris_tested - c0.tris_tested_shadow
        alg = {"traversal": (c.nodes_visited * 64 + c.tris_tested * 48) / n,
               "shade": (c.closest_hits * (12 + 72 + 176 + 640 + (160 if env_mis else 0)) + c.texture_taps * 16 + misses * 64) / n,
               "generate+resolve": float(s.width * s.height * 32 * spp)}
        split = {"closest_hit_rays": {"rays": (c.rays_primary + c.rays_bounce) / n, "nodes": (c.nodes_visited - c.nodes_visited_shadow) / n, "tris": (c.tris_tested - c.tris_tested_shadow) / n},
                 "shadow_rays": {"rays": c.rays_shadow / n, "nodes": c.nodes_visited_shadow / n, "tris": c.tris_tested_shadow / n},
                 "primary_rays": {"rays": c0.rays_primary / n, "nodes": primary_nodes / n, "tris": primary_tris / n}}
        # what the shade stage reads from tables it staged into LDS once per workgroup (material header + three slots, the 96-B
        # instance row, the three coarsest level pairs of the importance pyramid) instead of fetching per hit
        lds = c.closest_hits * (640 + 176 + (96 if env_mis else 0)) / n
        ms = {"traversal": stage_ms[1] + stage_ms[3], "shade": stage_ms[2], "generate+resolve": stage_ms[0] + stage_ms[4]}
        # The individual kernels of the traversal family: k_wf_trace = the primary rays (stage events of kind "trace"); k_wf_traverse = the
        # shadow rays of a bounce + the closest-hit rays of the next, fused (stage events of kind "shadow", which also hold the one
        # k_wf_shadow launch that ends a pt_trace: ~1 % of them)
        kernels = {"k_wf_trace": {"ms_per_launch": stage_ms[1], "algorithmic_bytes_per_launch": (primary_nodes * 64 + primary_tris * 48) / n, "launches_per_pt_trace": 1},
                   "k_wf_traverse": {"ms_per_launch": stage_ms[3], "algorithmic_bytes_per_launch": alg["traversal"] - (primary_nodes * 64 + primary_tris * 48) / n,
                                     "launches_per_pt_trace": int(settings.max_bounces), "note": "includes the pt_trace's last k_wf_shadow launch (no bounce rays left to fuse)"},
                   "k_wf_shade": {"ms_per_launch": stage_ms[2], "algorithmic_bytes_per_launch": alg["shade"], "launches_per_pt_trace": int(settings.max_bounces) + 1}}
        # Roofs, every one a figure of /opt/skills/guides/MI355X_MICROARCH.md (the builder's own probe ceilings ride beside them as extra keys):
        #   HBM 8.0 TB/s spec (6.29 achievable)                                         :36, section HBM
        #   L2-resident gather 16.8-18.8 TB/s, Infinity-Cache-resident gather 8.6 TB/s   section "Indexed rows: gather into LDS", table
        # The traversal kernels gather 64-B nodes and 48-B triangle packets that live in the L2s and the Infinity Cache (PMC: the fabric moves
        # a fifth of their algorithmic bytes), so HBM is not their roof: they are priced against the guide's L2 gather rate (lower figure).
        GUIDE = {"hbm_peak_GBps": 8000.0, "hbm_achievable_GBps": 6290.0, "l2_gather_GBps": [16800.0, 18800.0], "infinity_cache_gather_GBps": 8600.0,
                 "source": "MI355X_MICROARCH.md: chip table 'HBM3E peak BW', section 'HBM', section 'Indexed rows: gather into LDS'"}
        OWN = {"l2_resident_64B_node_gather_GBps": 12800.0, "infinity_cache_resident_64B_node_gather_GBps": 4200.0, "random_128B_lines_from_hbm_GBps": 6400.0,
               "source": "tools/probes/quadload_probe.hip, random_sector_probe.hip (profiles/r02e_microbenchmarks.txt): this pipeline's own access patterns; not the roofline's peak"}
        roofs = {"traversal": ("l2-gather", GUIDE["l2_gather_GBps"][0]), "shade": ("hbm", GUIDE["hbm_peak_GBps"]), "generate+resolve": ("hbm", GUIDE["hbm_peak_GBps"])}
        pmc, pmc_kernels, pmc_head = {}, {}, None
        pmc_file = os.path.join("profiles", "r03_pmc_per_kernel.json")
        if os.path.exists(os.path.join(ROOT, pmc_file)) and args.config == "sponza" and not (args.width or args.height or args.animate):
            try:
                pm = json.load(open(os.path.join(ROOT, pmc_file)))
                if pm.get("samples_per_launch", 1) == spp:
                    pmc, pmc_kernels, pmc_head = pm.get("stages", {}), pm.get("kernels", {}), pm.get("git_head")
            except Exception:
                pmc = {}

        def entry(bytes_, ms_, roof):
            gbs = bytes_ / max(ms_, 1e-9) / 1e6
            return {"ms_per_launch": round(ms_, 4), "algorithmic_bytes_per_launch": round(bytes_), "achieved_GBps": round(gbs, 1),
                    "bound": roof[0], "peak_GBps": roof[1], "frac": round(gbs / roof[1], 4), "frac_of_hbm_peak": round(gbs / GUIDE["hbm_peak_GBps"], 4)}

        stages = {}
        for k2 in alg:
            e = entry(alg[k2], ms[k2], roofs[k2])
            if k2 == "traversal":
                e["frac_of_l2_gather_upper"] = round(e["achieved_GBps"] / GUIDE["l2_gather_GBps"][1], 4)
                e["frac_of_infinity_cache_gather"] = round(e["achieved_GBps"] / GUIDE["infinity_cache_gather_GBps"], 4)
                e["split"] = {k3: {k4: round(v4, 1) for k4, v4 in v3.items()} for k3, v3 in split.items()}
            if k2 == "shade":
                e["lds_table_bytes_per_launch"] = round(lds)
                e["fetched_bytes_per_launch"] = round(alg[k2] - lds)
                e["fetched_GBps"] = round((alg[k2] - lds) / max(ms[k2], 1e-9) / 1e6, 1)
                e["frac_fetched_of_hbm_peak"] = round(e["fetched_GBps"] / GUIDE["hbm_peak_GBps"], 4)
            if k2 in pmc:
                e["pmc"] = pmc[k2]
                e["hbm_traffic_GBps"] = round(pmc[k2]["hbm_bytes_per_launch"] / max(ms[k2], 1e-9) / 1e6, 1)
                e["hbm_traffic_frac_of_hbm_peak"] = round(e["hbm_traffic_GBps"] / GUIDE["hbm_peak_GBps"], 4)
                e["traffic_over_algorithmic"] = round(pmc[k2]["hbm_bytes_per_launch"] / max(alg[k2] - (lds if k2 == "shade" else 0.0), 1.0), 3)
            stages[k2] = e
        kout = {}
        for kn, kv in kernels.items():
            e = entry(kv["algorithmic_bytes_per_launch"], kv["ms_per_launch"], roofs["shade" if kn == "k_wf_shade" else "traversal"])
            e["launches_per_pt_trace"] = kv["launches_per_pt_trace"]
            if "note" in kv:
                e["note"] = kv["note"]
            if kn in pmc_kernels:
                e["pmc"] = pmc_kernels[kn]
            kout[kn] = e
        # the single kernel with the largest share of a launch's GPU time
        dominant = max(kernels, key=lambda kn: kernels[kn]["ms_per_launch"])
        family = "shade" if dominant == "k_wf_shade" else "traversal"
        limiter = {"shade": "scattered cache lines (a gather is priced at its 128-B line) and ~7 k dependent vector instructions per hit at 2 waves/SIMD (DESIGN.md section 4)",
                   "traversal": "cache-gather rate, not HBM: nodes and triangle packets are L2 / Infinity-Cache resident (the fabric moves a fifth of the algorithmic bytes), "
                                "with ~160 vector instructions per node step that are not hidden (DESIGN.md section 4)"}
        total_ms = sum(ms.values())
        mean_ms = sum(kernel_ms) / len(kernel_ms)
        total_alg = sum(alg.values())
        d = kout[dominant]
        traffic = pmc_kernels.get(dominant, {}).get("hbm_bytes_per_launch") if pmc_kernels else None
        traffic_how = "profile"
        live = None
        if not args.no_pmc_live and binding is None and not (args.width or args.height or args.stage_blocks or args.cull_null_shadow):   # (the child runs the default command)
            live = pmc_live(list(kernels.keys()), spp, args.config)
            if live:
                traffic, traffic_how = live[dominant]["hbm_bytes_per_launch"], "live"
                for kn in kout:
                    kout[kn]["pmc_live"] = {k3: round(v3) for k3, v3 in live[kn].items()}
        result["roofline"] = {
            "kernel": dominant, "kernel_share_of_launch": round(kernels[dominant]["ms_per_launch"] / max(total_ms, 1e-9), 4),
            "kernel_family": {"traversal": "k_wf_traverse + k_wf_trace + k_wf_shadow (one traversal code path, trace_persistent)", "shade": "k_wf_shade"}[family],
            # `bound`: the traversal kernels are fed by the L2s and the Infinity Cache -- priced against 8 TB/s of HBM their fraction comes out
            # above 1, which means "cache-served", not "beyond the roof" -- so their peak is the guide's L2 gather rate; frac_of_hbm_peak and
            # frac_of_infinity_cache_gather are kept beside it, and hbm_traffic_* says what the HBM side actually moved (PMC)
            "bound": d["bound"], "achieved": d["achieved_GBps"], "peak": d["peak_GBps"], "unit": "GB/s", "frac": round(d["achieved_GBps"] / d["peak_GBps"], 5),
            "peak_source": GUIDE["source"], "guide_ceilings": GUIDE, "own_probe_ceilings": OWN,
            "frac_of_hbm_peak": d["frac_of_hbm_peak"],
            "frac_of_infinity_cache_gather": round(d["achieved_GBps"] / GUIDE["infinity_cache_gather_GBps"], 4),
            "ms_per_launch": d["ms_per_launch"], "algorithmic_bytes_per_launch": d["algorithmic_bytes_per_launch"],
            "one_launch": "one pt_trace = %d sample(s) per pixel; ms and bytes are sums over the kernel's %d launches inside it" % (spp, d["launches_per_pt_trace"]),
            "traffic": traffic,
            "traffic_frac_of_hbm_peak": round(traffic / max(d["ms_per_launch"], 1e-9) / 1e6 / GUIDE["hbm_peak_GBps"], 4) if traffic else None,
            "traffic_source": (("measured in this run: two rocprofv3 child runs of a short version of this command (1 warm-up + 3 steps, same scene and samples per launch), "
                                "--pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; (2 x FETCH_SIZE + WRITE_SIZE) x 1024, the gfx950 correction of MI355X_MICROARCH.md section HBM")
                               if traffic_how == "live" else
                               (pmc_file + " (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command: (2 x FETCH_SIZE + WRITE_SIZE) x 1024, the gfx950 correction of "
                                "MI355X_MICROARCH.md section HBM; collected at git head " + str(pmc_head) + ", not in this run)")) if traffic else None,
            "traffic_git_head": (None if traffic_how == "live" else pmc_head) if traffic else None,
            "traffic_measured_in_this_run": traffic_how == "live" if traffic else None,
            "limiter": limiter[family],
            "family": stages[family], "kernels": kout, "stages": stages,
            "pipeline": {"kernel_ms_mean_timed": round(mean_ms, 4), "stage_ms_sum_replay": round(total_ms, 4), "algorithmic_bytes_per_launch": round(total_alg),
                         "achieved_GBps": round(total_alg / (mean_ms * 1e-3) / 1e9, 1), "frac_of_hbm_peak": round(total_alg / (mean_ms * 1e-3) / 1e9 / 8000.0, 4),
                         "pmc": pmc.get("pipeline")},
            "nodes_per_ray": round(c.nodes_visited / max(c.rays, 1), 2), "tris_per_ray": round(c.tris_tested / max(c.rays, 1), 2), "rays_replay": int(c.rays),
            "bytes_per_ray": round(total_alg * n / max(c.rays, 1), 1)}

    # ---- CPU baseline: the oracle (port) on bounded samples of the same workload
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import pyoracle
        try:
            visible = len(os.sched_getaffinity(0))
        except Exception:
            visible = os.cpu_count() or 1
        cores = max(1, min(visible, 16))          # the GPU box's CPU share for one GPU is 16 cores, whatever the affinity mask shows
        o = pyoracle.Oracle()
        n_env, cube, pyr = r.env_read(h["env"]) if h["env"] is not None else (None, None, None)
        ho = s.upload(o, env_raw=(n_env, cube, pyr) if n_env else None)
        t1 = time.perf_counter(); o.build_accel(); build_wall = (time.perf_counter() - t1) * 1e3
        acc_ms, _ = o.timing()
        o.build_accel(cores)                   # the same tree (node for node, tested) on all the cores of the box's share: BASELINE.md section 3, leg B2
        acc_ms_all, _ = o.timing()
        sw, sh = s.width, s.height

        def cpu_trace(nthreads, budget_s, w, hgt):
            s.width, s.height = w, hgt
            img = np.zeros((hgt, w, 4), np.float32)
            o.counters()
            t1 = time.perf_counter()
            frames = 0
            while time.perf_counter() - t1 < budget_s and frames < 64:
                o.trace(settings, s.execute_params(frame=args.warmup + frames, env_handle=ho["env"]), img, nthreads=nthreads)
                frames += 1
            dt = time.perf_counter() - t1
            cc = o.counters()
            s.width, s.height = sw, sh
            return cc["rays"] / dt / 1e6, frames

        v_all, f_all = cpu_trace(cores, 10.0, max(sw // 8, 16), max(sh // 8, 16))       # 240x135 sample of the 1080p frame: same camera, same scene
        v_one, f_one = cpu_trace(1, 5.0, max(sw // 16, 16), max(sh // 16, 16))
        warm = {}
        for name, b in (("radix_tree", abi.BUILDER_LBVH), ("ploc", abi.BUILDER_PLOC), ("ploc_reinsert", abi.BUILDER_PLOC_REINSERT)):      # warm full builds of the bench scene, every builder
            r.set_accel_builder(b)
            ms_b = []
            for _ in range(4):
                r.request_rebuild(); r.build_accel(); torch.cuda.synchronize(); ms_b.append(r.stats().accel_ms)
            warm[name] = round(median(ms_b[1:]), 3)
        legs = {"B1_tracer_1_thread_Mrays": round(v_one, 4), "B1_tracer_all_cores_Mrays": round(v_all, 4),
                "B2_cpu_lbvh_build_1_thread_ms": round(acc_ms, 2), "B2_cpu_lbvh_build_all_cores_ms": round(acc_ms_all, 2), "B2_cpu_lbvh_build_cores": cores, "B2_hip_build_warm_ms": warm, "B2_hip_first_build_ms": round(accel_ms, 3), "B2_triangles": int(s.triangles),
                "B2_note": "CPU: oracle LBVH (Morton, stable sort, Karras splits, boxes on the way back up), on one thread and -- the same tree, chunk sorts + merges and subtrees per thread -- on all cores; the times include flattening the instances into world space; HIP: full on-device builds (PLOC + reinsertion is the default builder; "
                           "the first build also allocates); the reference's own BVH is built by the D3D12 driver"}
        # B2 refit + B4 skinning + B3 host work on the config-5 class scene (the dynamic path), GPU and CPU side by side
        try:
            legs.update(dynamic_legs(np, torch, scenes, abi, Renderer, pyoracle, local_rank))
        except Exception as e:        # noqa: BLE001 -- a baseline leg must never take the headline down
            legs["dynamic_legs_error"] = repr(e)
        result["cpu_baseline"] = {"value": round(v_all, 4), "unit": "Mrays/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(), "host_cpus_visible": visible,
                                  "value_1_thread": round(v_one, 4),
                                  "sample": "all cores: %d frames of the same scene/camera/settings at %dx%d (1/64 of the pixels); 1 thread: %d frames at %dx%d"
                                            % (f_all, max(sw // 8, 16), max(sh // 8, 16), f_one, max(sw // 16, 16), max(sh // 16, 16)),
                                  "legs": legs}

    if rank == 0:
        if args.save_image:
            from gltf_renderer_amd import gltf
            rgb, q = r.tonemap(out, want_rgba8=True)
            gltf.write_png(args.save_image, q, 3)
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def pmc_live(kernels, spp, config, timeout_s=240):
    """HBM-side traffic of the named kernels, measured IN THIS RUN the way MI355X_MICROARCH.md section HBM prescribes: FETCH_SIZE and WRITE_SIZE in
    SEPARATE rocprofv3 --pmc passes (child processes running a short version of the same command: same scene, same samples per launch, 1 warm-up +
    3 steps, nothing else), bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (the gfx950 correction), summed over a kernel's dispatches and divided by
    the pt_trace calls = per launch of the hot path.  Returns {kernel: {...}} or None if the profiler is not there or a pass fails."""
    import csv, glob, shutil, subprocess, tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None
    steps, warm = 3, 1
    cmd_tail = [sys.executable, os.path.abspath(__file__), "--steps", str(steps), "--warmup", str(warm), "--spp", str(spp), "--config", config,
                "--no-cpu-baseline", "--no-roofline", "--no-pmc-live"]
    env = dict(os.environ, TMPDIR="/tmp")
    res = {k: {} for k in kernels}
    tmp = tempfile.mkdtemp(prefix="mipt_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            p = subprocess.run([rocprof, "--pmc", counter, "--output-format", "csv", "-d", out, "--"] + cmd_tail, cwd="/tmp", env=env, capture_output=True, text=True,
                               timeout=timeout_s)
            if p.returncode != 0:
                return None
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return None
            for f in files:
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") != counter:
                        continue
                    for k in kernels:
                        if k in row.get("Kernel_Name", ""):
                            res[k][counter] = res[k].get(counter, 0.0) + float(row["Counter_Value"]) / float(steps + warm)
        for k, c in res.items():
            if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
                return None
            c["hbm_bytes_per_launch_raw"] = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
            c["hbm_bytes_per_launch"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        return res
    except Exception:          # noqa: BLE001 -- a profiler hiccup must not take the headline down
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def dynamic_legs(np, torch, scenes, abi, Renderer, pyoracle, device):
    """BASELINE.md section 3, legs B2 (refit), B3, B4 on the config-5 class scene: CPU restatement and HIP path side by side."""
    import tempfile
    legs = {}
    s = scenes.skinned_figure(256, 144)
    nv = s.skins[0]["mesh"].num_vertices
    # ---- B4: skinning.  CPU oracle (1 thread; Skin.cs.hlsl restated) vs k_skin / k_skin_mfma
    o = pyoracle.Oracle(); ho = s.upload(o)
    bo = scenes.SkinBinding(o, s, ho, 0, use_mfma=0)
    bones = scenes.bones_for_pose(bo.skin, np.eye(4), scenes.skinned_figure_pose(0.4))
    t0 = time.perf_counter(); reps = 0
    while time.perf_counter() - t0 < 1.0:
        o.skin_run(bo.params, bones); reps += 1
    cpu_skin_ms = (time.perf_counter() - t0) / reps * 1e3
    t0 = time.perf_counter(); o.build_accel(); cpu_build_ms = o.timing()[0]
    o.close()
    legs["B4_cpu_skin_1_thread_ms"] = round(cpu_skin_ms, 4)
    legs["B4_cpu_skin_Mverts_per_s"] = round(nv / cpu_skin_ms / 1e3, 2)
    r = Renderer(device=device); h = s.upload(r)
    for mfma in (0, 1):
        b = scenes.SkinBinding(r, s, h, 0, use_mfma=mfma)
        ms = []
        for k in range(12):
            b.pose(0.1 * k)
            torch.cuda.synchronize()
            ms.append(r.stats().skin_ms)
        legs["B4_hip_skin_%s_ms" % ("mfma" if mfma else "valu")] = round(median(ms[2:]), 4)
    legs["B4_vertices"] = int(nv)
    # ---- B2: refit vs full rebuild of the same dynamic scene, HIP; CPU LBVH build of it (the oracle has no refit: it rebuilds)
    b.pose(0.0); r.build_accel()
    refit, rebuild = [], []
    for k in range(10):
        b.pose(0.05 * k); r.build_accel(); torch.cuda.synchronize(); refit.append(r.stats().accel_ms)
    for k in range(6):
        b.pose(0.05 * k); r.request_rebuild(); r.build_accel(); torch.cuda.synchronize(); rebuild.append(r.stats().accel_ms)
    legs["B2_dynamic_scene_triangles"] = int(s.triangles)
    legs["B2_hip_refit_ms"] = round(median(refit[2:]), 4)
    legs["B2_hip_rebuild_ms"] = round(median(rebuild[1:]), 4)
    legs["B2_cpu_lbvh_build_dynamic_scene_ms"] = round(cpu_build_ms, 3)
    r.close()
    # ---- B3: the reference's per-frame host work restated (Gltf::Animate + CalculateGlobalTransforms, PerformSkinning's bone
    # matrices, GatherLights, GatherMaterials, BuildTlas' instance walk: Gltf.cpp:977-1041, Renderer.cpp:399-500, Pathtracer.cpp:185-257),
    # single thread as upstream, on the figure scene exported to .glb and loaded back by the C++ loader
    from gltf_renderer_amd.gltf import GltfScene
    from tests.scene_export import skinned_figure_to_builder
    with tempfile.TemporaryDirectory() as tmp:
        path = skinned_figure_to_builder(s).write_glb(os.path.join(tmp, "figure.glb"))
        r = Renderer(device=device)
        sc = GltfScene(path)
        sc.upload(r)
        host_us = []
        for k in range(60):
            t0 = time.perf_counter()
            sc.animate(0, (k % 60) / 30.0)
            sc.calculate_global_transforms(0)
            sc.frame(r, 0)
            host_us.append((time.perf_counter() - t0) * 1e6)
            if k % 8 == 7:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        legs["B3_host_work_per_frame_us"] = round(median(host_us[5:]), 1)
        legs["B3_note"] = "gs_animate + gs_calculate_global_transforms + gs_frame (C++ in libmipt.so, 1 thread, uploads staged asynchronously) on the %d-joint figure scene" % len(s.skins[0]["inverse_bind"])
        sc.unload(r); sc.close(); r.close()
    return legs


if __name__ == "__main__":
    main()
