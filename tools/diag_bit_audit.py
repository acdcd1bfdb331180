#!/usr/bin/env python3
"""Run ON THE GPU BOX: which of the reference's debug outputs (per-pixel deterministic quantities of the first path vertex) does the HIP path
reproduce BIT FOR BIT against the oracle, and which only to rounding?  usage: python tools/diag_bit_audit.py [sponza|test|grid|helmet|figure]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gltf_renderer_amd import abi, scenes
from gltf_renderer_amd.renderer import Renderer
from oracle import pyoracle
which = sys.argv[1] if len(sys.argv) > 1 else "sponza"
s = {"sponza": lambda: scenes.sponza_class(width=320, height=180, tex=64), "test": lambda: scenes.test_scene(160, 64), "grid": lambda: scenes.material_grid(256, seg=16),
     "helmet": lambda: scenes.helmet_class(width=320, height=180, subdiv=4, tex=256), "figure": lambda: scenes.skinned_figure(320, 180)}[which]()
r = Renderer(); hg = s.upload(r)
o = pyoracle.Oracle(); ho = s.upload(o, env_raw=r.env_read(hg["env"]) if hg["env"] is not None else None)
og = r.create_output(s.width, s.height); b = np.zeros((s.height, s.width, 4), np.float32)
for dbg in range(1, 28):
    st = abi.PtSettings.from_buffer_copy(bytes(s.settings)); st.debug_output = dbg; st.flags &= ~abi.FLAG_ACCUMULATE; st.use_frame_as_seed = 0; st.seed = 5
    r.trace(st, s.execute_params(frame=0, env_handle=hg["env"]), og)
    o.trace(st, s.execute_params(frame=0, env_handle=ho["env"]), b)
    a = r.readback(og)[..., :3]; bb = b[..., :3]
    diff = (a.view(np.uint32) != bb.view(np.uint32)).any(axis=2) & ~(np.isnan(a).any(axis=2) & np.isnan(bb).any(axis=2))
    rel = np.abs(a.astype(np.float64) - bb).max(axis=2) / np.maximum(np.abs(bb).max(axis=2), 1e-30)
    rel = np.where(np.isfinite(rel), rel, 0)
    print("%-28s pixels not bit-identical: %6d of %d   max relative difference %.2e" % (abi.DEBUG_OUTPUT_NAMES[dbg], int(diff.sum()), diff.size, float(rel.max())))
