"""Diagnostic (GPU box): which punctual light produces GPU-vs-oracle differences at the first hit (max_bounces 0)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from gltf_renderer_amd import abi, scenes
from gltf_renderer_amd.renderer import Renderer
import oracle.pyoracle as po

W, H = 320, 180
ROUGH = len(sys.argv) > 1 and sys.argv[1] == "rough"      # every material fully rough: is the difference the GGX peak's conditioning?
base = scenes.sponza_class(width=W, height=H, tex=64)
all_lights = list(base.lights)
for which in range(len(all_lights)):
    s = scenes.sponza_class(width=W, height=H, tex=64)
    s.lights = [all_lights[which]]
    if ROUGH:
        for m in s.materials: m.roughness_factor = 1.0; m.metallic_roughness.descriptor = -1
    r = Renderer(); hg = s.upload(r)
    o = po.Oracle(); ho = s.upload(o, env_raw=r.env_read(hg["env"]))
    for extra, name in ((0, "shadow rays"), (abi.FLAG_SHADOW_RAYS, "untraced")):
        st = abi.PtSettings.from_buffer_copy(bytes(s.settings)); st.max_bounces = 0; st.min_bounces = 0
        st.flags &= ~(abi.FLAG_ACCUMULATE | abi.FLAG_ENVIRONMENT_MIS | extra); st.use_frame_as_seed = 0; st.seed = 9
        og = r.create_output(W, H); b = np.zeros((H, W, 4), np.float32)
        r.trace(st, s.execute_params(0, env_handle=hg["env"]), og); o.trace(st, s.execute_params(0, env_handle=ho["env"]), b)
        a = r.readback(og)[..., :3].astype(np.float64); bb = b[..., :3].astype(np.float64)
        rel = np.abs(a - bb).max(axis=2) / np.maximum(np.abs(bb).max(axis=2), 1e-9)
        L = all_lights[which]
        print("light %d type %d  %-11s: pixels rel>1e-4 %.5f  max rel %.3e" % (which, L.type, name, (rel > 1e-4).mean(), rel.max()))
    r.close(); o.close()
