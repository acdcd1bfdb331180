#!/usr/bin/env python3
"""Run ON THE GPU BOX: at which bounce does a given pixel-sample of the Sponza-class scene part from the oracle?  usage: diag_config3_bounce.py frame x y"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gltf_renderer_amd import abi, scenes
from gltf_renderer_amd.renderer import Renderer
from oracle import pyoracle
f, x, y = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
s = scenes.sponza_class(width=320, height=180, tex=64)
r = Renderer(); hg = s.upload(r)
o = pyoracle.Oracle(); ho = s.upload(o, env_raw=r.env_read(hg["env"]))
og = r.create_output(s.width, s.height); b = np.zeros((s.height, s.width, 4), np.float32)
for variant in ("full", "no_lights", "no_env_mis", "no_shadow_rays"):
    for mb in range(0, 9):
        st = abi.PtSettings.from_buffer_copy(bytes(s.settings)); st.flags &= ~abi.FLAG_ACCUMULATE
        st.max_bounces = mb; st.min_bounces = min(st.min_bounces, mb)
        if variant == "no_lights": st.flags &= ~abi.FLAG_POINT_LIGHTS
        if variant == "no_env_mis": st.flags &= ~abi.FLAG_ENVIRONMENT_MIS
        if variant == "no_shadow_rays": st.flags &= ~abi.FLAG_SHADOW_RAYS
        r.trace(st, s.execute_params(frame=f, env_handle=hg["env"]), og)
        o.trace(st, s.execute_params(frame=f, env_handle=ho["env"]), b)
        a = r.readback(og)[y, x, :3]; bb = b[y, x, :3]
        print("%-14s max_bounces %d: gpu %s oracle %s  %s" % (variant, mb, np.array2string(a, precision=6), np.array2string(bb, precision=6), "" if np.allclose(a, bb, rtol=1e-4, atol=1e-7) else "<-- differ"))
