import sys; sys.path.insert(0, '.')
import numpy as np
from tests.test_gpu_round3 import _deep_chain_scene, copy_settings
from gltf_renderer_amd import abi
from gltf_renderer_amd.renderer import Renderer
for dups in (16, 4096, 65536):
    s, n = _deep_chain_scene(dups=dups)
    for b in (abi.BUILDER_LBVH, abi.BUILDER_PLOC_REINSERT):
        r = Renderer(); r.set_accel_builder(b); h = s.upload(r)
        r.enable_counters(True); r.reset_stats()
        st = copy_settings(s.settings); st.max_bounces = 0; st.min_bounces = 0; st.flags &= ~abi.FLAG_ACCUMULATE
        out = r.create_output(s.width, s.height)
        r.trace(st, s.execute_params(0), out)
        q = r.stats()
        print("dups %d builder %d: need %d cap %d deep %d; rays %d nodes/ray %.1f tris/ray %.1f bvh nodes %d" % (dups, b, q.bvh_stack_need, q.bvh_stack_capacity, q.deep_stack_pushes, q.rays, q.nodes_visited / max(q.rays, 1), q.tris_tested / max(q.rays, 1), q.bvh_nodes))
        r.close()
