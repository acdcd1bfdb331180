#!/usr/bin/env python3
"""Frame-time ablation on the Sponza-class scene: which part of the vertex costs what (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gltf_renderer_amd import scenes, abi, Renderer
s = scenes.sponza_class()
r = Renderer(); h = s.upload(r); r.build_accel()
out = r.create_output(s.width, s.height)
def run(name, st, frames=6):
    for f in range(2): r.trace(st, s.execute_params(f, env_handle=h["env"]), out)
    torch.cuda.synchronize(); r.reset_stats(); t0 = time.perf_counter()
    for f in range(frames): r.trace(st, s.execute_params(10 + f, env_handle=h["env"]), out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / frames
    c = r.stats()
    print("%-34s %7.3f ms  rays/frame %6.2fM  hits %5.2fM  Mrays/s %7.1f" % (name, dt * 1e3, c.rays / frames / 1e6, c.closest_hits / frames / 1e6, c.rays / frames / dt / 1e6))
base = abi.PtSettings.from_buffer_copy(bytes(s.settings))
run("baseline", base)
def mod(**kw):
    st = abi.PtSettings.from_buffer_copy(bytes(base))
    st.flags = (st.flags | kw.get("set", 0)) & ~kw.get("clr", 0)
    if "maxb" in kw: st.max_bounces = kw["maxb"]
    return st
run("diffuse_white", mod(set=abi.FLAG_MATERIAL_DIFFUSE_WHITE))
run("no env MIS (no env shadow rays)", mod(clr=abi.FLAG_ENVIRONMENT_MIS))
run("no point lights", mod(clr=abi.FLAG_POINT_LIGHTS))
run("no shadow rays (lights untraced)", mod(clr=abi.FLAG_SHADOW_RAYS))
run("no env MIS, no point lights", mod(clr=abi.FLAG_ENVIRONMENT_MIS | abi.FLAG_POINT_LIGHTS))
run("cosine only (no material MIS)", mod(clr=abi.FLAG_MATERIAL_MIS))
run("max_bounces 0", mod(maxb=0))
run("max_bounces 1", mod(maxb=1))
run("geometric normals", mod(set=abi.FLAG_MATERIAL_USE_GEOMETRIC_NORMALS))
st = mod(); st.debug_output = abi.DEBUG_OUTPUT_VERTEX_NORMAL
run("debug vertex normal (vertex fetch)", st)
st = mod(); st.debug_output = abi.DEBUG_OUTPUT_SHADING_NORMAL
run("debug shading normal (+textures)", st)
