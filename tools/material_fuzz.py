#!/usr/bin/env python3
"""Run ON THE GPU BOX: random MATERIALS and LIGHTS on the test scene (every mesh kind: textured, mirrored, non-indexed, no tangent space, MASK /
BLEND quads), GPU against the oracle pixel-sample by pixel-sample.  Every material's factors are redrawn (metalness, roughness, ior, specular,
clearcoat, anisotropy strength / rotation, sheen, transmission, emissive, alpha mode / cutoff, double-sidedness), each of the 15 texture slots is
bound or not with a random texture, sampler (wrap / mirror / clamp, point / linear), UV set and KHR_texture_transform; lights get random types,
positions, ranges and cone angles; the camera a random orbit pose and field of view (orthographic a quarter of the time).   usage: python tools/material_fuzz.py [trials] [seed]"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gltf_renderer_amd import abi, scenes
from gltf_renderer_amd.renderer import Renderer
import oracle.pyoracle as po


def randomize(s, rng):
    n_tex = len(s.textures)
    samplers = [0] + [s.add_sampler(int(rng.integers(0, 3)) + abi.ADDRESS_WRAP if False else int(rng.choice([abi.ADDRESS_WRAP, abi.ADDRESS_MIRROR, abi.ADDRESS_CLAMP])),
                                    int(rng.choice([abi.ADDRESS_WRAP, abi.ADDRESS_MIRROR, abi.ADDRESS_CLAMP])),
                                    int(rng.choice([abi.FILTER_POINT, abi.FILTER_LINEAR])), int(rng.choice([abi.FILTER_POINT, abi.FILTER_LINEAR]))) for _ in range(4)]
    u = lambda a=0.0, b=1.0: float(rng.uniform(a, b))
    for k, m in enumerate(s.materials):
        if k == 0: continue                                                    # index 0 stays the reference's default material
        m.flags = abi.MATERIAL_FLAG_DOUBLE_SIDED if rng.random() < 0.3 else 0
        m.alpha_mode = int(rng.choice([abi.ALPHA_MODE_OPAQUE, abi.ALPHA_MODE_OPAQUE, abi.ALPHA_MODE_MASK, abi.ALPHA_MODE_BLEND]))
        m.alpha_cutoff = u(0.2, 0.8) if m.alpha_mode == abi.ALPHA_MODE_MASK else 0.0                  # forced to 0 unless MASK (Renderer.h:145)
        m.metalness_factor = float(rng.choice([0.0, 1.0, u()])); m.roughness_factor = float(rng.choice([0.0, 1.0, u(), u(0, 0.1)]))
        m.base_color_factor[:] = (u(), u(), u(), float(rng.choice([1.0, u()])))
        m.emissive_factor[:] = (0, 0, 0) if rng.random() < 0.7 else (u(0, 3), u(0, 3), u(0, 3))
        m.ior = float(rng.choice([1.0, 1.33, 1.5, u(1.0, 2.5)])); m.normal_scale = u(0, 2)
        m.specular_factor = float(rng.choice([1.0, 0.0, u()])); m.specular_color_factor[:] = (u(), u(), u()) if rng.random() < 0.5 else (1, 1, 1)
        m.clearcoat_factor = float(rng.choice([0.0, 0.0, 1.0, u()])); m.clearcoat_roughness_factor = u(); m.clearcoat_normal_scale = u(0, 2)
        m.anisotropy_strength = float(rng.choice([0.0, 0.0, u(), 1.0])); m.anisotropy_rotation = u(-7, 7)
        m.sheen_color_factor[:] = (0, 0, 0) if rng.random() < 0.6 else (u(), u(), u()); m.sheen_roughness_factor = float(rng.choice([u(), 0.0, 1.0]))
        m.transmission_factor = float(rng.choice([0.0, 0.0, 1.0, u()]))
        for slot in abi.PtMaterial.TEXTURE_SLOTS:
            ts = getattr(m, slot)
            if rng.random() < 0.45:
                ts.descriptor = int(rng.integers(0, n_tex)); ts.sampler = int(rng.choice(samplers)); ts.tex_coord = int(rng.integers(0, 2))
                ts.rotation = 0.0 if rng.random() < 0.5 else u(-4, 4)
                ts.offset[:] = (0, 0) if rng.random() < 0.5 else (u(-2, 2), u(-2, 2)); ts.scale[:] = (1, 1) if rng.random() < 0.5 else (u(-3, 3), u(0.1, 4))
            else:
                ts.descriptor = -1
    # the camera: any orbit pose around the scene, any field of view; orthographic a quarter of the time
    from gltf_renderer_amd import camera
    s.world_to_view = camera.orbit_world_to_view((u(-0.5, 0.5), u(-0.5, 0.5), u(0.2, 1.0)), u(1.5, 6.0), u(-3.2, 3.2), u(-0.3, 1.3))
    s.y_fov = u(0.3, 2.6); s.ortho = (u(0.2, 1.0), u(0.2, 1.0)) if rng.random() < 0.25 else None
    for l in s.lights:
        l.type = int(rng.choice([abi.LIGHT_POINT, abi.LIGHT_SPOT, abi.LIGHT_DIRECTIONAL]))
        l.position[:] = (u(-3, 3), u(-3, 3), u(0.2, 4)); d = rng.standard_normal(3); l.direction[:] = d / np.linalg.norm(d)
        l.cutoff = float(rng.choice([0.0, u(1, 10)])); l.intensity = u(0.5, 30); l.color[:] = (u(), u(), u())
        l.inner_angle = u(0, 1.2); l.outer_angle = float(rng.choice([l.inner_angle, l.inner_angle + u(0, 0.5)]))     # equal angles: the 0.001 floor of the cone scale


if __name__ == "__main__":
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    rng = np.random.default_rng(seed); bad_total = total = 0; t0 = time.time()
    for t in range(trials):
        s = scenes.test_scene(144, 48, seed=3 + t % 3)
        randomize(s, rng)
        r = Renderer(); hg = s.upload(r)
        o = po.Oracle(); ho = s.upload(o, env_raw=r.env_read(hg["env"]))
        og = r.create_output(s.width, s.height); b = np.zeros((s.height, s.width, 4), np.float32)
        st = abi.PtSettings.from_buffer_copy(bytes(s.settings)); st.flags &= ~abi.FLAG_ACCUMULATE
        if rng.random() < 0.3: st.flags |= abi.FLAG_ALPHA_SHADOWS
        if rng.random() < 0.3: st.flags |= abi.FLAG_CULL_BACKFACE
        for frame in range(3):
            r.reset_stats(); o.counters()
            r.trace(st, s.execute_params(frame, env_handle=hg["env"]), og); o.trace(st, s.execute_params(frame, env_handle=ho["env"]), b)
            A = r.readback(og)[..., :3].astype(np.float64); B = b[..., :3].astype(np.float64)
            mism = int((np.isfinite(A).all(axis=2) != np.isfinite(B).all(axis=2)).sum())
            fin = np.isfinite(A).all(axis=2) & np.isfinite(B).all(axis=2)
            rel = np.where(fin, np.abs(A - B).max(axis=2) / np.maximum(np.abs(B).max(axis=2), 1e-4), 0)
            bad = int((rel > 1e-3).sum()) + mism; rg, ro = r.stats().rays, o.counters()["rays"]
            total += rel.size; bad_total += bad
            if bad or rg != ro:
                ys, xs = np.nonzero(rel > 1e-3)
                print("trial %d frame %d: %d pixel-samples beyond 1e-3 (%d non-finite mismatches), rays GPU %d oracle %d, worst %.2e at %s" % (t, frame, bad, mism, rg, ro, rel.max(), (xs[:3], ys[:3])), flush=True)
        r.close(); o.close()
    print("TOTAL: %d of %d pixel-samples beyond 1e-3 in %d random material / light sets x 3 frames (%.0f s)" % (bad_total, total, trials, time.time() - t0))
