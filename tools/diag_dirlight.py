"""Diagnostic (GPU box): pixels where the directional light's first-hit contribution differs between GPU and oracle."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from gltf_renderer_amd import abi, scenes
from gltf_renderer_amd.renderer import Renderer
import oracle.pyoracle as po

W, H = 320, 180
s = scenes.sponza_class(width=W, height=H, tex=64)
s.lights = [l for l in s.lights if l.type == 2]
r = Renderer(); hg = s.upload(r)
o = po.Oracle(); ho = s.upload(o, env_raw=r.env_read(hg["env"]))
def render(dbg=0, clear=0):
    st = abi.PtSettings.from_buffer_copy(bytes(s.settings)); st.max_bounces = 0; st.min_bounces = 0; st.debug_output = dbg
    st.flags &= ~(abi.FLAG_ACCUMULATE | abi.FLAG_ENVIRONMENT_MIS | abi.FLAG_SHADOW_RAYS | clear); st.use_frame_as_seed = 0; st.seed = 9
    og = r.create_output(W, H); b = np.zeros((H, W, 4), np.float32)
    r.trace(st, s.execute_params(0, env_handle=hg["env"]), og); o.trace(st, s.execute_params(0, env_handle=ho["env"]), b)
    return r.readback(og)[..., :3].astype(np.float64), b[..., :3].astype(np.float64)
a, b = render()
rel = np.abs(a - b).max(axis=2) / np.maximum(np.abs(b).max(axis=2), 1e-9)
ys, xs = np.nonzero(rel > 1e-3)
print("differing pixels:", len(ys), "of", W * H)
dbgs = {"shading_normal": abi.DEBUG_OUTPUT_SHADING_NORMAL, "vertex_normal": abi.DEBUG_OUTPUT_VERTEX_NORMAL, "roughness": abi.DEBUG_OUTPUT_ROUGHNESS,
        "metalness": abi.DEBUG_OUTPUT_METALNESS, "color": abi.DEBUG_OUTPUT_COLOR, "hit_kind": abi.DEBUG_OUTPUT_HIT_KIND, "clearcoat": abi.DEBUG_OUTPUT_CLEARCOAT,
        "transmissive": abi.DEBUG_OUTPUT_TRANSMISSIVE, "alpha": abi.DEBUG_OUTPUT_ALPHA}
maps = {k: render(v) for k, v in dbgs.items()}
a_g, b_g = render(clear=abi.FLAG_SHADING_NORMAL_ADAPTATION)
a_c, b_c = render(clear=abi.FLAG_MATERIAL_MIS)
for k in range(min(8, len(ys))):
    y, x = ys[k], xs[k]
    print("px (%d,%d) gpu %s oracle %s ratio %.4f | no-adapt ratio %.4f | cosine-only ratio %.4f" % (x, y, a[y, x], b[y, x], a[y, x].sum() / max(b[y, x].sum(), 1e-12),
          a_g[y, x].sum() / max(b_g[y, x].sum(), 1e-12), a_c[y, x].sum() / max(b_c[y, x].sum(), 1e-12)))
    for name, (ma, mb) in maps.items():
        print("      %-14s gpu %s oracle %s" % (name, np.round(ma[y, x], 5), np.round(mb[y, x], 5)))
