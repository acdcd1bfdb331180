#!/usr/bin/env python3
"""Run ON THE GPU BOX: where the first vertex's BSDF value is not bit-identical to the oracle's, what kind of surface is it?  Prints the
other debug outputs (all bit-identical on both sides) of a few such pixels and the statistics of the differing set.  usage: diag_bsdf_bits.py [sponza|grid]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gltf_renderer_amd import abi, scenes
from gltf_renderer_amd.renderer import Renderer
from oracle import pyoracle
which = sys.argv[1] if len(sys.argv) > 1 else "sponza"
s = {"sponza": lambda: scenes.sponza_class(width=320, height=180, tex=64), "grid": lambda: scenes.material_grid(256, seg=16)}[which]()
r = Renderer(); hg = s.upload(r)
o = pyoracle.Oracle(); ho = s.upload(o, env_raw=r.env_read(hg["env"]) if hg["env"] is not None else None)
og = r.create_output(s.width, s.height); b = np.zeros((s.height, s.width, 4), np.float32)
out = {}
for dbg in range(1, 28):
    st = abi.PtSettings.from_buffer_copy(bytes(s.settings)); st.debug_output = dbg; st.flags &= ~abi.FLAG_ACCUMULATE; st.use_frame_as_seed = 0; st.seed = 5
    r.trace(st, s.execute_params(frame=0, env_handle=hg["env"]), og); o.trace(st, s.execute_params(frame=0, env_handle=ho["env"]), b)
    out[abi.DEBUG_OUTPUT_NAMES[dbg]] = (r.readback(og)[..., :3].copy(), b[..., :3].copy())
a, bb = out["bounce_bsdf"]
diff = (a.view(np.uint32) != bb.view(np.uint32)).any(axis=2)
print("%d of %d pixels differ" % (diff.sum(), diff.size))
for name in ("metalness", "roughness", "specular", "clearcoat", "transmissive", "bounce_is_transmission", "hemisphere_view_side", "hit_kind", "alpha"):
    if name not in out: continue
    v = out[name][1][..., 0]
    print("  %-24s differing pixels: mean %.4f min %.4f max %.4f   | all pixels: mean %.4f" % (name, v[diff].mean(), v[diff].min(), v[diff].max(), v[np.isfinite(v)].mean()))
ys, xs = np.nonzero(diff)
for k in range(0, len(ys), max(1, len(ys) // 8)):
    y, x = ys[k], xs[k]
    print("pixel (%d,%d): bsdf gpu %s oracle %s" % (x, y, a[y, x], bb[y, x]))
    for name in ("metalness", "roughness", "specular", "specular_color", "clearcoat", "transmissive", "bounce_direction", "shading_normal", "bounce_pdf", "bounce_weight", "color"):
        if name in out: print("      %-18s %s" % (name, out[name][1][y, x]))
