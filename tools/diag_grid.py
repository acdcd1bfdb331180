#!/usr/bin/env python3
"""Run ON THE GPU BOX: where does the GPU / oracle difference of the config-4 material grid sit?  Per debug output and per grid cell
(6 x 6 spheres: rows 0-1 transmission, 2-3 clearcoat, 4 sheen, 5 anisotropy), signed and absolute tone-mapped differences."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gltf_renderer_amd import abi, scenes
from gltf_renderer_amd.renderer import Renderer
from oracle import pyoracle

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
s = scenes.material_grid(96, seg=12)
r = Renderer(); hg = s.upload(r)
o = pyoracle.Oracle(); ho = s.upload(o, env_raw=r.env_read(hg["env"]))
cp = lambda st: abi.PtSettings.from_buffer_copy(bytes(st))
for dbg in range(1, 28):
    st = cp(s.settings); st.debug_output = dbg; st.flags &= ~abi.FLAG_ACCUMULATE; st.use_frame_as_seed = 0; st.seed = 5
    og = r.create_output(s.width, s.height); b = np.zeros((s.height, s.width, 4), np.float32)
    r.trace(st, s.execute_params(0, env_handle=hg["env"]), og); o.trace(st, s.execute_params(0, env_handle=ho["env"]), b)
    a = r.readback(og)
    err = np.abs(a[..., :3] - b[..., :3]).max(axis=2)
    tol = 2e-4 + 2e-3 * np.abs(b[..., :3]).max(axis=2)
    print("debug %2d %-24s bad %.4f max %.3e" % (dbg, abi.DEBUG_OUTPUT_NAMES[dbg], float((err > tol).mean()), float(err.max())))
for mb in (0, 1, 2, 4, 16):
    st = cp(s.settings); st.max_bounces = mb; st.min_bounces = min(2, mb); st.reset = 1
    og = r.create_output(s.width, s.height); b = np.zeros((s.height, s.width, 4), np.float32)
    for f in range(spp):
        r.trace(st, s.execute_params(f, env_handle=hg["env"]), og); o.trace(st, s.execute_params(f, env_handle=ho["env"]), b); st.reset = 0
    ta, tb = r.tonemap(og).astype(np.float64), pyoracle.tonemap(b).astype(np.float64)
    ok = np.isfinite(ta).all(axis=2) & np.isfinite(tb).all(axis=2)
    d = np.where(ok[..., None], ta - tb, 0.0)
    e = np.sqrt((d ** 2).sum() / (np.where(ok[..., None], tb, 0) ** 2).sum())
    lin = r.readback(og)[..., :3].astype(np.float64) - b[..., :3]
    print("max_bounces %2d, %d spp: rel L2 %.3e, signed mean diff %.3e (tone-mapped), linear signed mean %.3e / mean %.3e" % (mb, spp, e, d.mean(), lin.mean(), b[..., :3].mean()))
    if mb == 16:
        sq = (d ** 2).sum(axis=2)
        H, W = sq.shape
        print("share of the squared error by 8 x 8 image block (rows top to bottom), percent:")
        for by in range(8):
            print(" ".join("%5.1f" % (100 * sq[by * H // 8:(by + 1) * H // 8, bx * W // 8:(bx + 1) * W // 8].sum() / sq.sum()) for bx in range(8)))
        print("signed mean diff by block x 1e4:")
        for by in range(8):
            print(" ".join("%6.1f" % (1e4 * d[by * H // 8:(by + 1) * H // 8, bx * W // 8:(bx + 1) * W // 8].mean()) for bx in range(8)))
