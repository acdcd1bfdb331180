#!/usr/bin/env python3
"""Render a .gltf / .glb file on an MI355X through the C-ABI only: loader (include/mipt_scene.h) -> path tracer (include/mipt.h).

  python tools/render_gltf.py scene.glb --env sky.hdr --spp 64 --size 1280 720 --out frame.png [--animation 0 --time 0.5]

Camera: an orbit camera fitted to the scene's bounds (the reference's default controller, CameraController.h:42-49);
settings: the application defaults (Main.cpp:462-474) with --bounces."""
import argparse
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--env", default="")
    ap.add_argument("--spp", type=int, default=32)
    ap.add_argument("--size", type=int, nargs=2, default=[1280, 720])
    ap.add_argument("--bounces", type=int, default=5)
    ap.add_argument("--animation", type=int, default=-1)
    ap.add_argument("--time", type=float, default=0.0)
    ap.add_argument("--azimuth", type=float, default=0.6)
    ap.add_argument("--inclination", type=float, default=-0.35)
    ap.add_argument("--out", default="frame.png")
    a = ap.parse_args()

    import torch
    from gltf_renderer_amd import abi, camera, gltf
    from gltf_renderer_amd.renderer import Renderer

    r = Renderer(0)
    sc = gltf.GltfScene(a.path)
    sc.upload(r)
    if a.animation >= 0:
        sc.animate(a.animation, a.time)
    sc.calculate_global_transforms(0)
    lights = sc.frame(r, 0)
    # bounds from the loaded streams and node transforms
    lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
    c = sc.counts()
    first_flat, f = {}, 0
    for m in range(c.meshes):
        first_flat[m] = f
        while f < c.primitives and sc.primitive(f)["mesh"] == m:
            f += 1
    for n in range(c.nodes):
        ni = sc.node(n)
        if ni.mesh < 0:
            continue
        g = np.array(ni.global_transform[:], np.float64).reshape(4, 4).T
        k = first_flat[ni.mesh]
        while k < c.primitives and sc.primitive(k)["mesh"] == ni.mesh:
            p = sc.primitive(k)["position"]
            if p is not None and len(p):
                w = p.astype(np.float64) @ g[:3, :3].T + g[:3, 3]
                lo, hi = np.minimum(lo, w.min(0)), np.maximum(hi, w.max(0))
            k += 1
    centre, radius = (lo + hi) / 2, float(np.linalg.norm(hi - lo)) / 2
    env = None
    if a.env:
        img, _ = gltf.load_rgb32f(a.env)
        env = r.env_create(np.ascontiguousarray(img))
    st = abi.PtSettings.app_defaults()
    st.max_bounces = a.bounces
    if env is None:
        st.flags &= ~(abi.FLAG_ENVIRONMENT_MAP | abi.FLAG_ENVIRONMENT_MIS)
        st.environment_color[:] = (0.6, 0.7, 0.9)
    r.set_bounce_limit(max(a.bounces, abi.REFERENCE_MAX_BOUNCES))
    w, h = a.size
    p = abi.PtExecuteParams()
    p.world_to_view[:] = camera.cm(camera.orbit_world_to_view(tuple(centre), 2.2 * radius, a.azimuth, a.inclination))
    p.view_to_clip[:] = camera.cm(camera.view_to_clip(w / h, math.radians(60), 0.01 * radius, 100.0 * radius))
    p.width, p.height, p.light_count = w, h, lights
    p.environment_map = -1 if env is None else env
    p.tile_rank, p.tile_rank_count = 0, 1
    out = r.create_output(w, h)
    for frame in range(a.spp):
        p.frame = frame
        r.trace(st, p, out)
    torch.cuda.synchronize()
    _, rgba8 = r.tonemap(out, want_rgba8=True)
    if a.out.lower().endswith(".exr"):
        gltf.write_exr(a.out, r.readback(out)[..., :3], half=True)       # linear radiance
    elif a.out.lower().endswith(".pfm"):
        gltf.write_pfm(a.out, r.readback(out)[..., :3])
    else:
        gltf.write_png(a.out, rgba8, 3)                                  # tone-mapped (AgX + sRGB), ToneMapper.ps.hlsl
    s = r.stats()
    print("%s: %d triangles, %d lights, %d spp, %.2f ms/frame, %.0f Mrays/s -> %s" % (a.path, s.bvh_triangles, lights, a.spp, s.trace_ms, s.rays / max(s.trace_ms, 1e-9) / 1e3 / max(a.spp, 1) * 1.0, a.out))


if __name__ == "__main__":
    main()
