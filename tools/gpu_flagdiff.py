#!/usr/bin/env python3
"""Per-pixel single-sample GPU-vs-oracle diff for one flag combination (diagnostics; run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gltf_renderer_amd import scenes, abi, Renderer
from oracle import pyoracle
set_f = int(sys.argv[1], 0); clr_f = int(sys.argv[2], 0); frames = int(sys.argv[3]) if len(sys.argv) > 3 else 4
s = scenes.test_scene(96, 64)
r = Renderer(); hg = s.upload(r)
o = pyoracle.Oracle(); ho = s.upload(o, env_raw=r.env_read(hg["env"]))
st = abi.PtSettings.from_buffer_copy(bytes(s.settings)); st.flags = ((st.flags | set_f) & ~clr_f) & ~abi.FLAG_ACCUMULATE
for f in range(frames):
    og = r.create_output(s.width, s.height); b = np.zeros((s.height, s.width, 4), np.float32)
    r.reset_stats(); o.counters()
    r.trace(st, s.execute_params(f, env_handle=hg["env"]), og); o.trace(st, s.execute_params(f, env_handle=ho["env"]), b)
    a = r.readback(og)
    d = np.abs(a[..., :3] - b[..., :3]).max(axis=2)
    rel = d / (np.abs(b[..., :3]).max(axis=2) + 1e-3)
    idx = np.argsort(rel.ravel())[::-1][:6]
    print("frame", f, "frac rel>1e-3:", float((rel > 1e-3).mean()), "rays", r.stats().rays, o.counters()["rays"])
    for i in idx:
        y, x = divmod(int(i), s.width)
        print("   px", x, y, "gpu", a[y, x, :3], "cpu", b[y, x, :3])
    for dbg in (abi.DEBUG_OUTPUT_HIT_KIND,):
        pass
# --- drill down on the worst pixel of the last frame with debug outputs
y, x = divmod(int(idx[0]), s.width)
for name, extra in (("show_nan", abi.FLAG_SHOW_NAN), ("show_inf", abi.FLAG_SHOW_INF)):
    st2 = abi.PtSettings.from_buffer_copy(bytes(st)); st2.flags |= extra
    og = r.create_output(s.width, s.height); r.trace(st2, s.execute_params(frames - 1, env_handle=hg["env"]), og)
    print(name, r.readback(og)[y, x, :3])
for dbg in (1, 9, 11, 12, 13, 14, 15, 18, 19, 20, 21, 22, 23, 24, 25, 26):
    st2 = abi.PtSettings.from_buffer_copy(bytes(st)); st2.debug_output = dbg
    og = r.create_output(s.width, s.height); b = np.zeros((s.height, s.width, 4), np.float32)
    r.trace(st2, s.execute_params(frames - 1, env_handle=hg["env"]), og); o.trace(st2, s.execute_params(frames - 1, env_handle=ho["env"]), b)
    print("dbg %-22s gpu %s cpu %s" % (abi.DEBUG_OUTPUT_NAMES[dbg], r.readback(og)[y, x, :3], b[y, x, :3]))
