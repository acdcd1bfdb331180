#!/usr/bin/env python3
"""Run ON THE GPU BOX: random flag / bounce / Russian-roulette combinations x the four stand-in scenes, single-sample frames, GPU against the
oracle pixel-sample by pixel-sample: how many differ by more than 1e-3, and are the ray counts equal?  A sweep for classes of disagreement the
fixed tests do not reach.   usage: python tools/flag_sweep.py [combinations] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gltf_renderer_amd import abi, scenes
from gltf_renderer_amd.renderer import Renderer
import oracle.pyoracle as po
n_combo = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
FLAGS = [abi.FLAG_CULL_BACKFACE, abi.FLAG_LUMINANCE_CLAMP, abi.FLAG_INDIRECT_ENVIRONMENT_ONLY, abi.FLAG_POINT_LIGHTS, abi.FLAG_SHADOW_RAYS, abi.FLAG_ALPHA_SHADOWS,
         abi.FLAG_ENVIRONMENT_MAP, abi.FLAG_ENVIRONMENT_MIS, abi.FLAG_MATERIAL_DIFFUSE_WHITE, abi.FLAG_MATERIAL_USE_GEOMETRIC_NORMALS, abi.FLAG_MATERIAL_MIS,
         abi.FLAG_SHOW_NAN, abi.FLAG_SHOW_INF, abi.FLAG_SHADING_NORMAL_ADAPTATION]
cases = [("test", scenes.test_scene(160, 64)), ("sponza", scenes.sponza_class(width=480, height=270, tex=128)), ("grid", scenes.material_grid(size=320, seg=16)),
         ("helmet", scenes.helmet_class(width=320, height=180, subdiv=4, tex=256))]
total = bad_total = 0; t0 = time.time()
for name, s in cases:
    r = Renderer(); hg = s.upload(r)
    o = po.Oracle(); ho = s.upload(o, env_raw=r.env_read(hg["env"]) if hg["env"] is not None else None)
    if s.bounce_limit != 5: r.set_bounce_limit(s.bounce_limit); o.set_bounce_limit(s.bounce_limit)
    og = r.create_output(s.width, s.height); b = np.zeros((s.height, s.width, 4), np.float32)
    for c in range(n_combo):
        st = abi.PtSettings.from_buffer_copy(bytes(s.settings))
        fl = abi.FLAG_NONE if rng.random() < 0.1 else 0
        for f in FLAGS:
            if rng.random() < (0.7 if f & abi.APP_DEFAULT_FLAGS else 0.3): fl |= f
        st.flags = fl
        st.max_bounces = int(rng.integers(0, s.bounce_limit + 1)); st.min_bounces = int(rng.integers(0, st.max_bounces + 1))
        st.min_russian_roulette_continue_prob = float(rng.choice([0.0, 0.1, 0.5])); st.max_russian_roulette_continue_prob = float(rng.choice([0.5, 0.9, 1.0]))
        st.luminance_clamp = float(rng.choice([1.0, 10.0, 100.0]))
        st.use_frame_as_seed = int(rng.integers(0, 2)); st.seed = int(rng.integers(0, 1 << 30))
        frame = int(rng.integers(0, 1000))
        r.reset_stats(); o.counters()
        r.trace(st, s.execute_params(frame, env_handle=hg["env"]), og); o.trace(st, s.execute_params(frame, env_handle=ho["env"]), b)
        A = r.readback(og)[..., :3].astype(np.float64); B = b[..., :3].astype(np.float64)
        fin = np.isfinite(A).all(axis=2) & np.isfinite(B).all(axis=2)
        nonfinite_mismatch = int((np.isfinite(A).all(axis=2) != np.isfinite(B).all(axis=2)).sum())
        rel = np.where(fin, np.abs(A - B).max(axis=2) / np.maximum(np.abs(B).max(axis=2), 1e-4), 0)
        bad = int((rel > 1e-3).sum()) + nonfinite_mismatch
        rg, ro = r.stats().rays, o.counters()["rays"]
        total += rel.size; bad_total += bad
        if bad or rg != ro:
            print("%-7s flags 0x%04x bounces %d/%d rr %.1f-%.1f clamp %g frame %d: %d pixel-samples beyond 1e-3 (non-finite mismatches %d), rays GPU %d oracle %d, worst %.2e"
                  % (name, fl, st.min_bounces, st.max_bounces, st.min_russian_roulette_continue_prob, st.max_russian_roulette_continue_prob, st.luminance_clamp, frame, bad, nonfinite_mismatch, rg, ro, rel.max()), flush=True)
    r.close(); o.close()
    print("%s: %d combinations done (%.0f s)" % (name, n_combo, time.time() - t0), flush=True)
print("TOTAL: %d of %d pixel-samples beyond 1e-3" % (bad_total, total))
