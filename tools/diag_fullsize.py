"""Diagnostic (GPU box): per-pixel differences GPU vs oracle on the 257 k-triangle scene at a reduced frame, by bounce limit."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from gltf_renderer_amd import abi, scenes
from gltf_renderer_amd.renderer import Renderer
import oracle.pyoracle as po

s = scenes.sponza_class(width=320, height=180, tex=64)
r = Renderer(); hg = s.upload(r)
o = po.Oracle(); ho = s.upload(o, env_raw=r.env_read(hg["env"]))
for mb, flags_clear, name in ((0, 0, "mb0"), (0, abi.FLAG_POINT_LIGHTS, "mb0 no lights"), (0, abi.FLAG_ENVIRONMENT_MIS, "mb0 no envmis"), (1, 0, "mb1"), (4, 0, "mb4")):
    st = abi.PtSettings.from_buffer_copy(bytes(s.settings)); st.max_bounces = mb; st.min_bounces = min(st.min_bounces, mb); st.flags &= ~(abi.FLAG_ACCUMULATE | flags_clear)
    st.use_frame_as_seed = 0; st.seed = 9
    og = r.create_output(s.width, s.height); b = np.zeros((s.height, s.width, 4), np.float32)
    r.trace(st, s.execute_params(0, env_handle=hg["env"]), og); o.trace(st, s.execute_params(0, env_handle=ho["env"]), b)
    a = r.readback(og)[..., :3].astype(np.float64); b = b[..., :3].astype(np.float64)
    d = np.abs(a - b).max(axis=2); m = np.maximum(np.abs(b).max(axis=2), 1e-6)
    rel = d / m
    print("%-14s pixels rel>1e-4: %.4f  rel>1e-2: %.4f  rel>0.5: %.4f   median rel %.2e   sum|a| %.4e sum|b| %.4e" % (
        name, (rel > 1e-4).mean(), (rel > 1e-2).mean(), (rel > 0.5).mean(), np.median(rel), np.abs(a).sum(), np.abs(b).sum()))
    ys, xs = np.nonzero((rel > 1e-3) & (rel < 0.1))
    for k in range(min(4, len(ys))):
        print("    px (%d,%d) gpu %s oracle %s" % (xs[k], ys[k], a[ys[k], xs[k]], b[ys[k], xs[k]]))
