#!/usr/bin/env python3
"""Run ON THE GPU BOX: nodes per ray and Mrays/s of the reinsertion builder over passes x minimum gain (MIPT_REINSERT_* are read at
pt_create).   usage: python tools/reinsert_sweep.py [sponza|grid|helmet]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gltf_renderer_amd import abi, scenes
from gltf_renderer_amd.renderer import Renderer
which = sys.argv[1] if len(sys.argv) > 1 else "sponza"
s = {"sponza": scenes.sponza_class, "helmet": scenes.helmet_class, "grid": scenes.material_grid}[which]()
for passes in (0, 4, 6, 8, 12, 16, 24):
    for eps in ("1e-3",):
        os.environ["MIPT_REINSERT_PASSES"] = str(passes); os.environ["MIPT_REINSERT_MIN_GAIN"] = eps
        r = Renderer(); r.set_accel_builder(abi.BUILDER_PLOC_REINSERT)
        h = s.upload(r); r.build_accel(); r.request_rebuild(); r.build_accel(); torch.cuda.synchronize()
        q = r.stats()
        r.set_samples_per_trace(8)
        st = abi.PtSettings.from_buffer_copy(bytes(s.settings)); st.reset = 1
        out = r.create_output(s.width, s.height)
        for f in range(3):
            r.trace(st, s.execute_params(f * 8, env_handle=h["env"]), out); st.reset = 0
        torch.cuda.synchronize(); r.reset_stats()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for f in range(3, 9):
            r.trace(st, s.execute_params(f * 8, env_handle=h["env"]), out)
        e.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(e) / 6; rays = r.stats().rays / 6
        r.enable_counters(True); r.reset_stats(); r.set_samples_per_trace(1)
        r.trace(st, s.execute_params(99, env_handle=h["env"]), out)
        c = r.stats()
        print("%s passes %2d min gain %-5s build %.2f ms  stack need %d  nodes/ray %.2f  %.0f Mrays/s" % (which, passes, eps, q.accel_ms, q.bvh_stack_need, c.nodes_visited / c.rays, rays / ms / 1e3), flush=True)
        r.close()
