#!/usr/bin/env python3
"""Run ON THE GPU BOX with a -DPT_UTIL_PROBE build of the library (tools/build_variant.sh util -DPT_UTIL_PROBE): lane utilisation of
the traversal loop by phase, closest-hit and occlusion rays apart.   usage: python tools/util_probe.py [sponza|grid] [lib]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2: os.environ["MIPT_LIBRARY"] = sys.argv[2]
import torch
from gltf_renderer_amd import abi, scenes, renderer
from gltf_renderer_amd.renderer import Renderer
which = sys.argv[1] if len(sys.argv) > 1 else "sponza"
s = {"sponza": scenes.sponza_class, "grid": scenes.material_grid}[which]()
r = Renderer(); h = s.upload(r); r.build_accel(); r.set_samples_per_trace(8)
L = r.L
buf = (C.c_ulonglong * 16)()
st = abi.PtSettings.from_buffer_copy(bytes(s.settings)); st.reset = 1
out = r.create_output(s.width, s.height)
r.trace(st, s.execute_params(0, env_handle=h["env"]), out); st.reset = 0
L.pt_debug_read_util(buf, 1)
r.enable_counters(True); r.reset_stats()
r.trace(st, s.execute_params(8, env_handle=h["env"]), out)
L.pt_debug_read_util(buf, 0)
c = r.stats()
for mode, name in ((0, "closest-hit rays"), (1, "occlusion rays")):
    v = [int(buf[mode * 8 + k]) for k in range(7)]
    print("%s: node iterations %d (lanes stepping %.1f of 64, lanes holding a ray %.1f), leaf iterations %d (lanes testing %.1f), refills %d (%.1f rays each)"
          % (name, v[0], v[1] / max(v[0], 1), v[6] / max(v[0], 1), v[2], v[3] / max(v[2], 1), v[4], v[5] / max(v[4], 1)))
print("rays %d, node visits %d (%d by occlusion rays), triangle tests %d" % (c.rays, c.nodes_visited, c.nodes_visited_shadow, c.tris_tested))
