#!/usr/bin/env python3
"""Headline benchmark: Mrays/s + ms/frame of the path-tracing hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched under
torch.distributed.run, one rank per GPU.  A "step" = one pt_trace of the whole frame (1 sample per
pixel, BASELINE.json metric) on the Sponza-class stand-in (configs[2]: 1920x1080, 8 bounces + RR,
punctual lights + env MIS).  With N ranks the frame is sharded by 16x16 pixel tile (tile t -> rank
t % N), every rank renders its tiles and ONE exchange per frame assembles the image on rank 0: by
default each rank sends only its own tiles point to point over its direct xGMI link (--exchange gather),
or an RCCL reduce(sum) of the zeroed full-size image (--exchange reduce); tiles are disjoint, so both
give the same bits.

Samples per step.  One pt_trace carries a SAMPLE BATCH (pt_set_samples_per_trace): S samples per pixel in
one set of kernel launches, bit-identical to S consecutive reference frames (tested).  An offline path
tracer accumulates hundreds of samples, and a batch keeps 256 CUs full through the thin late bounces:
1080p on one MI355X goes from 2.56 Grays/s at S = 1 to 3.29 at S = 8 (3.48 at 32).  Default S = 8 per GPU;
`config.ms_per_1spp_frame` is the step time / S and `config.ms_single_sample_launch` the latency of an
S = 1 launch, measured beside it.

Scaling (N > 1).  Default "weak": a step accumulates 8 x N samples per pixel of the tile-sharded frame, so
every rank keeps the work of a 1-GPU step (1/N of the tiles x 8N samples) in ONE batch - a 1/N tile shard
of a single sample cannot fill a 256-CU GPU (measured: 0.99 Grays/s per GPU at N = 8), a batch can (2.5) -
then one exchange of the accumulated tiles.  `--scaling strong` keeps 8 samples per pixel per step split N
ways (total work fixed).  The JSON line says which.

Timed region: inputs (scene, BVH, textures, env maps) are resident in HBM; K steps bracketed by
barrier + torch.cuda.synchronize on both sides; time = max over ranks.  value = rays traced by all
ranks in the K steps / that time.  Rays are counted by the kernel itself (every traversal started).

Extra objects on the JSON line:
  roofline     - the wavefront pipeline of one pt_trace (one launch = one step = S samples): algorithmic bytes per
                 launch (counted by an untimed instrumented replay of the same K frames: nodes*64 +
                 tris*48 + hits*S_hit + taps*16 + env loads + 32 B/pixel) / mean per-frame kernel time
                 measured live with HIP events on the launch stream; peak = 8 TB/s HBM; traffic =
                 PMC-measured HBM bytes per launch from profiles/pmc_traffic.json (same command), or null.
  cpu_baseline - the CPU oracle (kind "port": the reference has no CPU tracer) on a bounded sample of
                 the same workload, all host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="sponza", choices=["sponza", "helmet", "grid", "figure", "test"])
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--save-image", default="")
    ap.add_argument("--mode", default="wavefront", choices=["wavefront", "megakernel"])
    ap.add_argument("--stage-blocks", type=int, default=0)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo + --single-device rehearses N ranks on a 1-GPU box")
    ap.add_argument("--single-device", action="store_true", help="rehearsal only: every rank uses cuda:0 (RCCL needs one device per rank)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = every step accumulates N samples per pixel of the tile-sharded frame in one sample batch (per-GPU work fixed); "
                         "strong = every step is one sample per pixel split N ways (total work fixed)")
    ap.add_argument("--spp", type=int, default=0,
                    help="samples per pixel per step (sample batch, pt_set_samples_per_trace); default 8 on one GPU, 8 x N with --scaling weak (max 64)")
    ap.add_argument("--exchange", default="gather", choices=["gather", "reduce"], help="per-frame assembly on rank 0: own-tile gather (default) or full-image reduce")
    ap.add_argument("--cull-null-shadow", action="store_true", help="experiment: pt_set_null_shadow_culling(1) -- same image, fewer shadow rays than the reference traces (off for the headline line)")
    ap.add_argument("--animate", action="store_true", help="config 5 (--config figure): skin -> BVH rebuild -> trace every step, accumulation reset each frame")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.single_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)

    from gltf_renderer_amd import scenes, abi
    from gltf_renderer_amd.renderer import Renderer
    from gltf_renderer_amd.sharding import TileExchange, reduce_frame

    t_setup = time.time()
    if args.config == "sponza":
        s = scenes.sponza_class()
    elif args.config == "helmet":
        s = scenes.helmet_class()
    elif args.config == "grid":
        s = scenes.material_grid()
    elif args.config == "figure":
        s = scenes.skinned_figure()
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
            raise SystemExit("--animate needs a scene with a skinned mesh (--config figure)")
        binding = scenes.SkinBinding(r, s, h, 0, use_mfma=1)      # every rank skins and rebuilds itself
        binding.pose(0.0)
    r.build_accel()
    settings = s.settings
    SPP_PER_GPU = 8                           # an offline renderer accumulates many samples: 8 per launch amortise the stage tails
    spp = args.spp if args.spp > 0 else min(SPP_PER_GPU * (world if args.scaling == "weak" else 1), 64)
    if args.animate:
        spp = 1                               # a playing animation resets accumulation every frame (Main.cpp:521-523)
    r.set_samples_per_trace(spp)              # one launch carries the step's samples: a 1/N tile shard still fills the GPU
    r.set_null_shadow_culling(args.cull_null_shadow)
    out = r.create_output(s.width, s.height)
    torch.cuda.synchronize()
    accel_ms = r.stats().accel_ms
    t_setup = time.time() - t_setup

    # The one exchange per frame (sharding.py): "gather" = every rank sends only its own tiles to rank 0 point to point over
    # its direct xGMI link; "reduce" = RCCL reduce(sum) of the zeroed full-size image.  Both assemble the same bits.
    exchange = {"mode": args.exchange if world > 1 else "none"}
    xch = TileExchange(s.width, s.height, world, "cuda" if args.backend == "nccl" else "cpu") if world > 1 else None

    def reduce_image(img):
        if args.backend == "gloo":          # rehearsal path: gloo moves host tensors
            host = img.cpu()
            if exchange["mode"] == "gather":
                xch.gather_frame(host, rank)
            else:
                reduce_frame(host, world)
            if rank == 0:
                img.copy_(host)
        elif exchange["mode"] == "gather":
            xch.gather_frame(img, rank)
        else:
            reduce_frame(img, world)        # one RCCL reduce(sum) of the accumulation buffer over xGMI

    def clear_for_exchange():
        if exchange["mode"] == "reduce":    # the sum needs zeros outside this rank's tiles; the gather does not read them
            out.zero_()

    def animate(frame):
        if binding is not None:           # Main.cpp:521-523: a playing animation resets accumulation every frame
            binding.pose((frame % 60) / 30.0)
            r.build_accel()
            settings.reset = 1

    def step(frame):
        animate(frame)
        if world > 1:
            clear_for_exchange()
            settings.reset = 1          # each step is a fresh 1-spp frame when sharded (the exchange assembles disjoint tiles)
        p = s.execute_params(frame=frame * spp, tile_rank=rank, tile_rank_count=world, env_handle=h["env"])
        r.trace(settings, p, out)
        if world > 1:
            reduce_image(out)

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world > 1 and exchange["mode"] == "gather":
        try:                                  # probe once: a backend without gather falls back to the reduce on every rank alike
            step(0)
            torch.cuda.synchronize()
        except Exception as e:               # noqa: BLE001
            if rank == 0:
                print("gather exchange unavailable (%s); using reduce" % e, file=sys.stderr)
            exchange["mode"] = "reduce"
    for f in range(args.warmup):
        step(f)
    sync_all()
    r.reset_stats()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    sync_all()
    t0 = time.perf_counter()
    for k in range(args.steps):
        frame = args.warmup + k
        animate(frame)
        if world > 1:
            clear_for_exchange()
            settings.reset = 1
        p = s.execute_params(frame=frame * spp, tile_rank=rank, tile_rank_count=world, env_handle=h["env"])
        ev[k][0].record()
        r.trace(settings, p, out)
        ev[k][1].record()
        if world > 1:
            reduce_image(out)
    sync_all()
    elapsed = time.perf_counter() - t0
    st = r.stats()
    kernel_ms = [a.elapsed_time(b) for a, b in ev]
    rays_local = int(st.rays)

    stat_dev = "cuda" if args.backend == "nccl" else "cpu"
    el = torch.tensor([elapsed], dtype=torch.float64, device=stat_dev)
    ry = torch.tensor([rays_local], dtype=torch.float64, device=stat_dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(ry, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    rays_total = float(ry.item())

    result = None
    if rank == 0:
        mrays = rays_total / elapsed / 1e6
        result = {
            "metric": "Mrays/sec + ms/frame @1920x1080, 8-bounce, 1/2/4/8 MI355X",
            "value": round(mrays, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1000.0, 4), "higher_is_better": True,
            "scaling": "weak" if (args.scaling == "weak" and args.spp <= 0) else "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s %dx%d %dspp/step, max_bounces %d (limit %d), min_bounces %d, RR %.1f-%.1f, %d triangles / %d instances / %d textures / %d lights, env-map MIS, flags 0x%x"
                       % (s.name, s.width, s.height, spp, settings.max_bounces, s.bounce_limit, settings.min_bounces,
                          settings.min_russian_roulette_continue_prob, settings.max_russian_roulette_continue_prob, s.triangles,
                          len(s.instances), len(s.textures), len(s.lights), settings.flags),
                       "parallelism": ("tile-shard x%d + 1 RCCL %s/frame" % (world, "tile gather to rank 0 (point to point)" if exchange["mode"] == "gather" else "reduce(sum)"))
                                      if world > 1 else "single GPU",
                       "null_shadow_culling": bool(args.cull_null_shadow),
                       "samples_per_step": spp, "ms_per_1spp_frame": round(elapsed / args.steps / spp * 1000.0, 4),
                       "rays_per_frame": round(rays_total / args.steps / spp, 1), "bvh_build_ms": round(accel_ms, 3), "scene_setup_s": round(t_setup, 1)},
        }

    # ---- latency of a single-sample launch (one reference frame), untimed, beside the batched throughput
    if rank == 0 and world == 1 and spp > 1 and binding is None and not args.no_roofline:     # (--no-roofline = profiling runs: timed launches only)
        r.set_samples_per_trace(1)
        lat = []
        for k in range(6):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            r.trace(settings, s.execute_params(frame=100000 + k, env_handle=h["env"]), out)
            b.record()
            torch.cuda.synchronize()
            lat.append(a.elapsed_time(b))
        result["config"]["ms_single_sample_launch"] = round(sorted(lat[1:])[len(lat[1:]) // 2], 4)
        r.set_samples_per_trace(spp)

    # ---- config 5: skin / BVH rebuild / trace split, measured on an untimed synchronised replay of a few frames
    if rank == 0 and binding is not None:
        parts = {"skin_ms": [], "accel_ms": [], "trace_ms": []}
        for k in range(min(args.steps, 8)):
            animate(args.warmup + k)
            r.trace(settings, s.execute_params(frame=(args.warmup + k) * spp, tile_rank=rank, tile_rank_count=world, env_handle=h["env"]), out)
            torch.cuda.synchronize()
            st2 = r.stats()
            parts["skin_ms"].append(st2.skin_ms); parts["accel_ms"].append(st2.accel_ms); parts["trace_ms"].append(st2.trace_ms)
        result["config"]["dynamic_ms"] = {k2: round(sum(v) / len(v), 4) for k2, v in parts.items()}
        result["config"]["parallelism"] += "; skin (MFMA joint blend) + full LBVH rebuild + 1 spp trace per step"

    # ---- roofline: instrumented untimed replay of the same frames (rank 0, N = 1 only)
    if rank == 0 and world == 1 and not args.no_roofline:
        r.enable_counters(True)
        settings2 = abi.PtSettings.from_buffer_copy(bytes(settings))
        out2 = r.create_output(s.width, s.height)
        torch.cuda.synchronize()
        r.reset_stats()
        for k in range(args.steps):
            animate(args.warmup + k)
            if binding is not None:
                settings2.reset = 1
            p = s.execute_params(frame=(args.warmup + k) * spp, env_handle=h["env"])
            r.trace(settings2, p, out2)
        c = r.stats()
        r.enable_counters(False)
        n = float(args.steps)
        env_mis = bool(settings.flags & abi.FLAG_ENVIRONMENT_MIS) and bool(settings.flags & abi.FLAG_ENVIRONMENT_MAP)
        # bytes/ray = N_node*node_size + N_tri*48 + [closest hits] S_hit + 32/R per pixel-sample.  Node size is
        # 64 B (the quantised 4-wide node of this build; the survey's figure).
        s_hit = 12 + 3 * (12 + 4 + 8) + 176 + 640            # indices + 3 vertices + instance row + material
        alg = (c.nodes_visited * 64 + c.tris_tested * 48 + c.closest_hits * s_hit + c.texture_taps * 16
               + (c.closest_hits * 40 * 4 if env_mis else 0) + (c.rays_primary + c.rays_bounce - c.closest_hits) * 64) / n + s.width * s.height * 32 * spp
        mean_ms = sum(kernel_ms) / len(kernel_ms)
        achieved = alg / (mean_ms * 1e-3) / 1e9
        traffic = None
        pj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pj) and args.config == "sponza" and not (args.width or args.height or args.animate) and args.mode == "wavefront":
            try:
                pm = json.load(open(pj))
                if pm.get("samples_per_launch", 1) == spp:         # the PMC passes were taken at this batch size
                    traffic = pm.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        kname = ("pt_megakernel" if args.mode == "megakernel" else
                 "pt_trace wavefront pipeline (k_wf_generate + (k_wf_trace, k_wf_shade, k_wf_shadow) x (max_bounces+1) + k_wf_resolve; "
                 "one launch = one pt_trace = %d sample(s) per pixel)" % spp)
        result["roofline"] = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s",
                              "frac": round(achieved / 8000.0, 5), "traffic": traffic,
                              "algorithmic_bytes_per_launch": round(alg), "kernel_ms_mean": round(mean_ms, 4),
                              "nodes_per_ray": round(c.nodes_visited / max(c.rays, 1), 2), "tris_per_ray": round(c.tris_tested / max(c.rays, 1), 2),
                              "rays_replay": int(c.rays),
                              "note": "latency/issue-bound gather workload: BVH + geometry are L2 / Infinity-Cache resident, so measured HBM traffic is far below the algorithmic bytes (DESIGN.md)"}

    # ---- CPU baseline: oracle (port) on a bounded sample of the same workload
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import pyoracle
        try:
            cores = len(os.sched_getaffinity(0))
        except Exception:
            cores = os.cpu_count() or 1
        cores = max(1, min(cores, 16))       # the GPU box's CPU share for one GPU
        o = pyoracle.Oracle()
        n_env, cube, pyr = r.env_read(h["env"]) if h["env"] is not None else (None, None, None)
        ho = s.upload(o, env_raw=(n_env, cube, pyr) if n_env else None)
        o.build_accel()
        acc_ms, _ = o.timing()
        sw, sh = s.width, s.height
        s.width, s.height = max(sw // 8, 16), max(sh // 8, 16)       # 240x135 sample of the 1080p frame: same camera, same scene
        img = np.zeros((s.height, s.width, 4), np.float32)
        o.counters()
        t1 = time.perf_counter()
        frames = 0
        while time.perf_counter() - t1 < 10.0 and frames < 64:
            o.trace(settings, s.execute_params(frame=args.warmup + frames, env_handle=ho["env"]), img, nthreads=cores)
            frames += 1
        dt = time.perf_counter() - t1
        cc = o.counters()
        s.width, s.height = sw, sh
        result["cpu_baseline"] = {"value": round(cc["rays"] / dt / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
                                  "sample": "%d frames of the same scene/camera/settings at %dx%d (1/64 of the pixels), oracle LBVH build %.0f ms (1 thread)"
                                  % (frames, max(sw // 8, 16), max(sh // 8, 16), acc_ms)}

    if rank == 0:
        if args.save_image:
            from gltf_renderer_amd import gltf
            rgb, q = r.tonemap(out, want_rgba8=True)
            gltf.write_png(args.save_image, q, 3)
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
