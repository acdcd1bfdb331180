#!/usr/bin/env python3
"""Writes tests/golden/test_scene_32_spp4.npy: a 32x32, 4-sample radiance tile of scenes.test_scene rendered by the
CPU oracle (oracle/).  It is a regression pin of the oracle itself, not reference output."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gltf_renderer_amd import scenes
from oracle import pyoracle

s = scenes.test_scene(32, 16)
o = pyoracle.Oracle()
h = s.upload(o)
out = np.zeros((s.height, s.width, 4), np.float32)
for f in range(4):
    o.trace(s.settings, s.execute_params(frame=f, env_handle=h["env"]), out, nthreads=1)
np.save(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "test_scene_32_spp4.npy"), out)
print("mean", out[..., :3].mean())
