#!/usr/bin/env python3
"""Quick GPU-vs-oracle comparison on the small all-feature scene (run on the GPU box)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gltf_renderer_amd import scenes, abi, Renderer
from oracle import pyoracle

size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
s = scenes.test_scene(size, 64)
r = Renderer()
r.enable_counters(True)
hg = s.upload(r)
n, cube, pyr = r.env_read(hg["env"])
o = pyoracle.Oracle()
ho = s.upload(o, env_raw=(n, cube, pyr))
# env preprocessing parity (oracle's own K10-K13 vs GPU's)
o2 = pyoracle.Oracle(); e2 = o2.env_create(s.env_image); n2, cube2, pyr2 = o2.env_read(e2)
cg = np.vectorize(pyoracle.lib().orc_half_to_float)(cube[..., :3].astype(np.uint16)).astype(np.float64) if False else None
print("env cube size", n, n2, "pyramid rel err", float(np.abs(pyr - pyr2).max() / np.abs(pyr2).max()), "cube bits equal frac", float((cube == cube2).mean()))
res = {}
for dbg in range(1, 28):
    st = abi.PtSettings.from_buffer_copy(bytes(s.settings)); st.debug_output = dbg; st.flags &= ~abi.FLAG_ACCUMULATE
    st.use_frame_as_seed = 0; st.seed = 5
    og = r.create_output(s.width, s.height)
    r.trace(st, s.execute_params(0, env_handle=hg["env"]), og)
    a = r.readback(og)
    b = np.zeros((s.height, s.width, 4), np.float32)
    o.trace(st, s.execute_params(0, env_handle=ho["env"]), b)
    d = np.abs(a[..., :3] - b[..., :3]).max(axis=2)
    bad = float((d > 1e-3).mean())
    res[abi.DEBUG_OUTPUT_NAMES[dbg]] = (float(np.median(d)), float(d.max()), bad)
    print("dbg %-24s median %.2e max %.2e frac>1e-3 %.5f" % (abi.DEBUG_OUTPUT_NAMES[dbg], np.median(d), d.max(), bad))
# radiance
st = abi.PtSettings.from_buffer_copy(bytes(s.settings))
og = r.create_output(s.width, s.height)
b = np.zeros((s.height, s.width, 4), np.float32)
cg = {"rays": 0}
o.counters()
r.reset_stats()
t0 = time.time()
for f in range(spp):
    r.trace(st, s.execute_params(f, env_handle=hg["env"]), og)
stt = r.stats(); cg["rays"] = stt.rays
tg = time.time() - t0
for f in range(spp):
    o.trace(st, s.execute_params(f, env_handle=ho["env"]), b)
co = o.counters()
a = r.readback(og)
ta = r.tonemap(og); tb = pyoracle.tonemap(b)
rel = float(np.sqrt(((ta - tb) ** 2).sum() / (tb ** 2).sum()))
rel_lin = float(np.sqrt(((a[..., :3] - b[..., :3]) ** 2).sum() / (b[..., :3] ** 2).sum()))
pp = np.abs(ta - tb).max(axis=2)
print("radiance spp=%d  relL2(tonemapped)=%.3e relL2(linear)=%.3e  frac px diff>1e-2: %.5f  rays gpu %d oracle %d" % (spp, rel, rel_lin, float((pp > 1e-2).mean()), cg["rays"], co["rays"]))
print("last frame stats: trace_ms %.3f nodes %d tris %d hits %d taps %d" % (stt.trace_ms, stt.nodes_visited, stt.tris_tested, stt.closest_hits, stt.texture_taps))
os.makedirs("gpurun_out", exist_ok=True)
from PIL import Image
Image.fromarray((np.clip(ta, 0, 1) * 255).astype(np.uint8)).save("gpurun_out/check_gpu.png")
Image.fromarray((np.clip(tb, 0, 1) * 255).astype(np.uint8)).save("gpurun_out/check_oracle.png")
