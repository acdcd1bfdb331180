#!/usr/bin/env python3
"""Run ON THE GPU BOX: are two builds of the library bit-identical in their images?  Renders a few scenes with each (one subprocess per
build, MIPT_LIBRARY) and compares the accumulation buffers bit for bit.   usage: python tools/compare_builds.py <variant A | base> <variant B | base>"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if len(sys.argv) > 1 and sys.argv[1] == "--render":
    sys.path.insert(0, ROOT)
    from gltf_renderer_amd import scenes
    from gltf_renderer_amd.renderer import Renderer
    out_prefix = sys.argv[2]
    cases = [("test", scenes.test_scene(160, 64), 6), ("sponza", scenes.sponza_class(width=480, height=270, tex=256), 4),
             ("grid", scenes.material_grid(size=256, seg=16), 4), ("helmet", scenes.helmet_class(width=320, height=180, subdiv=4, tex=256), 4)]
    for name, s, frames in cases:
        r = Renderer(); h = s.upload(r)
        if s.bounce_limit != 5: r.set_bounce_limit(s.bounce_limit)
        out = r.create_output(s.width, s.height)
        for f in range(frames):
            r.trace(s.settings, s.execute_params(frame=f, env_handle=h["env"]), out)
        np.save(out_prefix + name + ".npy", r.readback(out)); r.close()
    sys.exit(0)

def lib(v): return None if v == "base" else os.path.join(ROOT, "variants", "libmipt_%s.so" % v)
a, b = sys.argv[1], sys.argv[2]
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for v in (a, b):
    env = dict(os.environ)
    if lib(v): env["MIPT_LIBRARY"] = lib(v)
    else: env.pop("MIPT_LIBRARY", None)
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "--render", os.path.join(ROOT, "gpurun_out", "cmp_%s_" % v)], env=env)
for name in ("test", "sponza", "grid", "helmet"):
    x = np.load(os.path.join(ROOT, "gpurun_out", "cmp_%s_%s.npy" % (a, name))); y = np.load(os.path.join(ROOT, "gpurun_out", "cmp_%s_%s.npy" % (b, name)))
    d = (x.view(np.uint32) != y.view(np.uint32)).any(axis=2)
    rel = float(np.sqrt(((x.astype(np.float64) - y) ** 2)[np.isfinite(x) & np.isfinite(y)].sum() / max((y.astype(np.float64) ** 2)[np.isfinite(y)].sum(), 1e-30)))
    print("%-8s %dx%d: %d of %d pixels differ, rel L2 of the linear images %.3e" % (name, x.shape[1], x.shape[0], int(d.sum()), d.size, rel))
