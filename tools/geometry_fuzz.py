#!/usr/bin/env python3
"""Run ON THE GPU BOX: random TRIANGLE SOUPS, the product's traversal (pt_debug_intersect) against the oracle's, ray by ray and bit for bit.
Each trial: 1-4 instances (random rotation / non-uniform scale / mirroring / translation up to 1e4 from the coordinate origin) of random soups
(50-4000 triangles, sizes over six orders of magnitude, some degenerate -- repeated or collinear vertices --, some exact coplanar duplicates so
that equal-distance ties occur, some axis-aligned sheets whose boxes have no thickness); all three builders in turn; rays between random points
of the scene's box, along the axes, from points ON triangles, grazing; closest-hit with and without culling and occlusion.  A sample of the rays
also goes through the oracle's exhaustive search.   usage: python tools/geometry_fuzz.py [trials] [seed]"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
from gltf_renderer_amd import abi, scenes, meshgen, camera
from gltf_renderer_amd.renderer import Renderer
import oracle.pyoracle as po
from ray_hook import gpu_intersect, dxr_flags, RF_CULL_BACK, RF_CULL_FRONT, RF_ACCEPT_FIRST
f32 = np.float32


def random_scene(rng):
    s = scenes.SceneData("fuzz")
    s.materials.append(scenes.material(flags=abi.MATERIAL_FLAG_DOUBLE_SIDED))        # material 1: culling disabled
    for inst in range(int(rng.integers(1, 5))):
        n = int(rng.integers(50, 4000)); size = 10.0 ** rng.uniform(-3, 1)
        c = rng.uniform(-1, 1, (n, 1, 3)) * 10.0 ** rng.uniform(-1, 1)
        tri = (c + rng.standard_normal((n, 3, 3)) * size * 10.0 ** rng.uniform(-2, 0, (n, 1, 1))).astype(f32)
        k = n // 10
        tri[:k, 1] = tri[:k, 0]                                                      # repeated vertex
        tri[k:2 * k, 2] = (tri[k:2 * k, 0] + tri[k:2 * k, 1]) * f32(0.5)             # collinear
        tri[2 * k:3 * k] = tri[3 * k:4 * k]                                           # exact duplicates: equal-distance ties
        tri[4 * k:5 * k, :, int(rng.integers(0, 3))] = f32(rng.uniform(-1, 1))        # an axis-aligned sheet: boxes without thickness
        m = meshgen.Mesh(tri.reshape(-1, 3), None if rng.random() < 0.5 else np.arange(3 * n))
        R = np.linalg.qr(rng.standard_normal((3, 3)))[0]
        if rng.random() < 0.3: R = np.eye(3)                                          # keep the sheet axis-aligned in world space
        S = np.diag(rng.uniform(0.2, 3, 3) * rng.choice([1, 1, -1], 3))
        T = np.eye(4); T[:3, :3] = R @ S; T[:3, 3] = rng.uniform(-1, 1, 3) * float(rng.choice([0, 1, 100, 1e4]))
        s.add_mesh(m, T, int(rng.integers(0, 2)))
    return s


def world_triangles(s):
    """[n, 3, 3] float64: every triangle of the scene in world space (for aiming rays)."""
    out = []
    for mesh, T, _ in s.mesh_records:
        P = mesh.positions.astype(np.float64) @ T[:3, :3].T + T[:3, 3]
        out.append((P if mesh.indices is None else P[mesh.indices]).reshape(-1, 3, 3))
    return np.concatenate(out)


def random_rays(rng, o, tris, n):
    """Half of the rays are aimed at a random point of a random triangle (from a random point of its instance's neighbourhood or of the whole
    scene), half run between random points of the scene's box."""
    lo, hi = tris.reshape(-1, 3).min(0), tris.reshape(-1, 3).max(0)
    ext = hi - lo
    p = (lo + rng.random((n, 3)) * ext).astype(f32); q = (lo + rng.random((n, 3)) * ext).astype(f32)
    k = n // 2
    tk = tris[rng.integers(0, len(tris), k)]
    w = rng.dirichlet((1, 1, 1), k)
    q[:k] = (tk * w[:, :, None]).sum(axis=1).astype(f32)
    near = rng.random(k) < 0.5                                             # from nearby: a few triangle sizes away
    size = np.linalg.norm(tk.max(axis=1) - tk.min(axis=1), axis=1, keepdims=True) + 1e-6
    p[:k][near] = (q[:k][near] + rng.standard_normal((int(near.sum()), 3)) * size[near] * 10.0 ** rng.uniform(0, 2, (int(near.sum()), 1))).astype(f32)
    d = q - p; d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-30)
    rays = np.zeros((n, 8), f32); rays[:, 0:3] = p; rays[:, 4:7] = d; rays[:, 7] = (f32(4) * np.linalg.norm(q - p, axis=1)).astype(f32)
    h = o.intersect_many(rays)
    on = h[:, 0] > 0
    surf = rays[on].copy(); surf[:, 0:3] = (rays[on, 0:3] + h[on, 1:2] * rays[on, 4:7]).astype(f32)          # origins ON triangles
    dn = rng.standard_normal((len(surf), 3)); surf[:, 4:7] = (dn / np.linalg.norm(dn, axis=1, keepdims=True)).astype(f32)
    ax = rays[: n // 4].copy(); ax[:, 4:7] = np.eye(3, dtype=f32)[rng.integers(0, 3, len(ax))] * rng.choice(f32([-1, 1]), len(ax))[:, None]
    short = rays[: n // 4].copy(); short[:, 7] *= rng.random(len(short)).astype(f32) * f32(0.05); short[:, 3] = short[:, 7] * f32(0.1)     # tmin > 0
    return np.concatenate([rays, surf, ax, short])


if __name__ == "__main__":
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
    n_rays = n_bad = n_hit = n_closest = 0; t0 = time.time()
    for t in range(trials):
        s = random_scene(rng)
        r = Renderer(); s.upload(r); r.set_accel_builder(int(t % 3))
        o = po.Oracle(); s.upload(o)
        rays = random_rays(rng, o, world_triangles(s), 20000)
        for flags, mode in ((0, 0), (RF_CULL_BACK, 0), (RF_CULL_FRONT, 0), (RF_ACCEPT_FIRST, 1)):
            g = gpu_intersect(r, rays, flags, mode); c = o.intersect_many(rays, dxr_flags(flags), mode)
            same = np.array_equal(g[:, 0], c[:, 0]) if mode == 1 else np.array_equal(g[:, :7].view(np.uint32), c[:, :7].view(np.uint32))
            n_rays += len(rays)
            if mode == 0 and flags == 0: n_hit += int(c[:, 0].sum()); n_closest += len(rays)
            if not same:
                d = np.nonzero((g[:, :7].view(np.uint32) != c[:, :7].view(np.uint32)).any(axis=1) if mode == 0 else g[:, 0] != c[:, 0])[0]
                n_bad += len(d)
                print("trial %d (builder %d, %d triangles, flags %d mode %d): %d of %d rays differ; first: ray %s gpu %s oracle %s" % (t, t % 3, s.triangles, flags, mode, len(d), len(rays), rays[d[0]], g[d[0]], c[d[0]]), flush=True)
        sub = rays[rng.integers(0, len(rays), 1500)]
        o.set_brute_force(True); b = o.intersect_many(sub, 0, 0); o.set_brute_force(False)
        if not np.array_equal(b.view(np.uint32), o.intersect_many(sub, 0, 0).view(np.uint32)):
            n_bad += 1; print("trial %d: the oracle's tree and its exhaustive search differ" % t, flush=True)
        r.close(); o.close()
    print("TOTAL: %d of %d ray queries differ in %d random scenes; %.0f %% of the unculled closest-hit rays hit something (%.0f s)" % (n_bad, n_rays, trials, 100.0 * n_hit / max(n_closest, 1), time.time() - t0))
