"""Diagnostic (GPU box): cycles per section of the shade stage's closest-hit function, from a library built with -DPT_TIMING
(variants/libmipt_timing.so = the library with pt_wavefront.hip compiled -DPT_TIMING).  Prints the share of each section.
build:  cd gltf_renderer_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -DPT_TIMING -c pt_wavefront.hip -o /tmp/wf_t.o &&
        hipcc -shared -fPIC --offload-arch=gfx950 -o ../../variants/libmipt_timing.so /tmp/wf_t.o mipt_api.o pt_kernel.o accel.o envmap.o skin_tonemap.o host/*.o"""
import ctypes as C, os, shutil, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ["MIPT_LIBRARY"] = os.path.join(ROOT, "variants", "libmipt_timing.so")     # built by tools/build_variant.sh timing -DPT_TIMING
import torch
from gltf_renderer_amd import scenes, renderer
from gltf_renderer_amd.renderer import Renderer
s = scenes.sponza_class(); r = Renderer(device=0); h = s.upload(r); r.build_accel()
r.set_samples_per_trace(8)
out = r.create_output(s.width, s.height)
L = renderer.load_library()
buf = (C.c_ulonglong * 12)()
for f in range(2): r.trace(s.settings, s.execute_params(frame=8 * f, env_handle=h["env"]), out)
L.pt_debug_read_timing(buf, 1)
for f in range(2, 5): r.trace(s.settings, s.execute_params(frame=8 * f, env_handle=h["env"]), out)
L.pt_debug_read_timing(buf, 0)
names = ["packet + instance + vertex attributes", "offsets + get_surface (textures)", "lobe probabilities + emissive", "environment NEE: BSDF evaluation, MIS", "light NEE", "BSDF sample + Russian roulette",
         "environment NEE: random numbers + descent", "environment NEE: direction + cube sample"]
tot = sum(buf[k] for k in range(8)); n = buf[11]
print("wave-hits %d, cycles per wave-hit %.0f" % (n, tot / max(n, 1)))
for k in range(8):
    print("  %-40s %6.1f %%   %8.0f cycles" % (names[k], 100.0 * buf[k] / tot, buf[k] / max(n, 1)))
