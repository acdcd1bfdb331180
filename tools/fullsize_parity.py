#!/usr/bin/env python3
"""Run ON THE GPU BOX: the north_star contract at the full size of a BASELINE config AND an accumulated sample count -- N samples of the whole
frame on both sides (the oracle on the box's host cores: ~2.7 s per 1080p sample of config 3), tone-mapped relative L2 after 1, 4, 16, ... N
samples, ray counts, pixels beyond thresholds.  Too slow for the test suite (tests/test_gpu_round3.py holds the one-sample version); the
output is kept in profiles/.   usage: python tools/fullsize_parity.py [sponza|helmet|grid|figure] [samples]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gltf_renderer_amd import abi, scenes
from gltf_renderer_amd.renderer import Renderer
import oracle.pyoracle as po

which = sys.argv[1] if len(sys.argv) > 1 else "sponza"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
s = {"sponza": scenes.sponza_class, "helmet": scenes.helmet_class, "grid": scenes.material_grid, "figure": scenes.skinned_figure}[which]()
r = Renderer(); hg = s.upload(r)
o = po.Oracle(); ho = s.upload(o, env_raw=r.env_read(hg["env"]) if hg["env"] is not None else None)
if which == "figure":
    for backend, h in ((r, hg), (o, ho)):
        scenes.SkinBinding(backend, s, h, 0, 0).pose(0.55)
st = abi.PtSettings.from_buffer_copy(bytes(s.settings)); st.reset = 1
og = r.create_output(s.width, s.height); b = np.zeros((s.height, s.width, 4), np.float32)
r.set_samples_per_trace(1)
r.reset_stats(); o.counters(); oracle_rays = 0          # o.counters() reads AND clears: accumulated here
t0 = time.time()
print("%s %dx%d, max_bounces %d, %d triangles: GPU against the oracle, sample by sample" % (s.name, s.width, s.height, s.settings.max_bounces, s.triangles), flush=True)
for f in range(N):
    r.trace(st, s.execute_params(f, env_handle=hg["env"]), og); o.trace(st, s.execute_params(f, env_handle=ho["env"]), b); st.reset = 0
    if (f + 1) in (1, 4, 16, 64, 96, 128, 160, 192, 224, 256, 384, 512) or f + 1 == N:
        ta, tb = r.tonemap(og), po.tonemap(b)
        ok = np.isfinite(ta).all(axis=2) & np.isfinite(tb).all(axis=2)
        e = float(np.sqrt(((ta[ok].astype(np.float64) - tb[ok]) ** 2).sum() / (tb[ok].astype(np.float64) ** 2).sum()))
        A = r.readback(og)[..., :3].astype(np.float64); B = b[..., :3].astype(np.float64)
        rel = np.abs(A - B).max(axis=2) / np.maximum(np.abs(B).max(axis=2), 1e-4)
        sg = r.stats(); oracle_rays += o.counters()["rays"]
        rad_bits = int((r.readback(og)[..., :3].view(np.uint32) != b[..., :3].view(np.uint32)).any(axis=2).sum())
        tm_bits = int((ta.view(np.uint32) != tb.view(np.uint32)).any(axis=2).sum())
        print("  %3d samples: tone-mapped rel L2 %.3e (contract 1e-3); pixels beyond 1e-3 / 1e-2 of the oracle's radiance: %d / %d of %d; pixels not bit-identical: radiance %d, tone-mapped %d; rays GPU %d, oracle %d (%s); %.0f s"
              % (f + 1, e, int((rel > 1e-3).sum()), int((rel > 1e-2).sum()), rel.size, rad_bits, tm_bits, sg.rays, oracle_rays, "equal" if sg.rays == oracle_rays else "DIFFERENT by %d" % (sg.rays - oracle_rays), time.time() - t0), flush=True)
r.close(); o.close()
