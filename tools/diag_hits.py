"""Diagnostic (GPU box): primary-hit attribute mismatches on the 257 k-triangle scene: GPU BVH vs oracle BVH vs oracle brute force."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from gltf_renderer_amd import abi, scenes
from gltf_renderer_amd.renderer import Renderer
import oracle.pyoracle as po

W, H = 160, 90
s = scenes.sponza_class(width=W, height=H, tex=64)
r = Renderer(); hg = s.upload(r)
o = po.Oracle(); ho = s.upload(o, env_raw=r.env_read(hg["env"]))
for dbg in (abi.DEBUG_OUTPUT_TEXCOORD_0, abi.DEBUG_OUTPUT_VERTEX_NORMAL):
    st = abi.PtSettings.from_buffer_copy(bytes(s.settings)); st.debug_output = dbg; st.flags &= ~abi.FLAG_ACCUMULATE; st.use_frame_as_seed = 0; st.seed = 3
    og = r.create_output(W, H); r.trace(st, s.execute_params(0, env_handle=hg["env"]), og); a = r.readback(og)[..., :3]
    b = np.zeros((H, W, 4), np.float32); o.set_brute_force(False); o.trace(st, s.execute_params(0, env_handle=ho["env"]), b)
    t = time.time(); c = np.zeros((H, W, 4), np.float32); o.set_brute_force(True); o.trace(st, s.execute_params(0, env_handle=ho["env"]), c); t = time.time() - t
    def mm(x, y): return float((np.abs(x - y[..., :3]).max(axis=2) > 1e-4).mean())
    print("%s: gpu vs oracle-bvh %.5f   gpu vs brute %.5f   oracle-bvh vs brute %.5f   (brute force %.1f s)" % (abi.DEBUG_OUTPUT_NAMES[dbg], mm(a, b), mm(a, c), mm(b[..., :3], c), t))
    d = np.abs(a - c[..., :3]).max(axis=2)
    ys, xs = np.nonzero(d > 1e-4)
    for k in range(min(6, len(ys))):
        print("    px (%d,%d) gpu %s brute %s bvh %s" % (xs[k], ys[k], a[ys[k], xs[k]], c[ys[k], xs[k], :3], b[ys[k], xs[k], :3]))
