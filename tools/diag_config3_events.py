#!/usr/bin/env python3
"""Run ON THE GPU BOX: which pixel-samples of the Sponza-class scene (320x180, its own 8 bounces) still differ from the oracle, frame by frame
(no accumulation), and by how much.   usage: python tools/diag_config3_events.py [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gltf_renderer_amd import abi, scenes
from gltf_renderer_amd.renderer import Renderer
from oracle import pyoracle
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 64
s = scenes.sponza_class(width=320, height=180, tex=64)
r = Renderer(); hg = s.upload(r)
o = pyoracle.Oracle(); ho = s.upload(o, env_raw=r.env_read(hg["env"]))
st = abi.PtSettings.from_buffer_copy(bytes(s.settings)); st.flags &= ~abi.FLAG_ACCUMULATE
og = r.create_output(s.width, s.height); b = np.zeros((s.height, s.width, 4), np.float32)
tot = 0
for f in range(frames):
    r.trace(st, s.execute_params(frame=f, env_handle=hg["env"]), og)
    o.trace(st, s.execute_params(frame=f, env_handle=ho["env"]), b)
    a = r.readback(og)[..., :3].astype(np.float64); bb = b[..., :3].astype(np.float64)
    rel = np.abs(a - bb).max(axis=2) / np.maximum(np.abs(bb).max(axis=2), 1e-6)
    bad = np.argwhere(rel > 1e-3)
    tot += len(bad)
    for y, x in bad[:6]:
        print("frame %2d pixel (%3d,%3d): gpu %s oracle %s rel %.3g" % (f, x, y, np.array2string(a[y, x], precision=5), np.array2string(bb[y, x], precision=5), rel[y, x]))
print("pixel-samples beyond 1e-3: %d of %d" % (tot, frames * s.width * s.height))
