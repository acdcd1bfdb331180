"""Test helper: the product's traversal on caller-supplied rays (pt_debug_intersect, a test hook of libmipt.so that is not part of include/mipt.h)."""
import ctypes as C
import numpy as np

RF_CULL_BACK, RF_CULL_FRONT, RF_FORCE_NON_OPAQUE, RF_ACCEPT_FIRST = 1, 2, 4, 8        # csrc/pt_traverse.h


def dxr_flags(rf):
    """The same flags in the D3D12 RAY_FLAG_* encoding the oracle's traversal takes (oracle.cpp)."""
    return (0x10 if rf & RF_CULL_BACK else 0) | (0x20 if rf & RF_CULL_FRONT else 0) | (0x2 if rf & RF_FORCE_NON_OPAQUE else 0) | (0x4 if rf & RF_ACCEPT_FIRST else 0)


def gpu_intersect(renderer, rays, ray_flags=0, mode=0):
    """rays [n, 8] float32 (origin, tmin, direction, tmax) -> [n, 8] float32 (committed, t, u, v, instance, primitive, front, transmission)."""
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
    out = np.zeros((len(rays), 8), np.float32)
    f = renderer.L.pt_debug_intersect
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
    f.restype = C.c_int
    rc = f(renderer.h, rays.ctypes.data_as(C.c_void_p), len(rays), ray_flags, mode, out.ctypes.data_as(C.c_void_p))
    if rc != 0: raise RuntimeError("pt_debug_intersect: %d %s" % (rc, renderer.last_error() if hasattr(renderer, "last_error") else ""))
    return out


def surface_rays(o, s, n, seed):
    """Rays as the path tracer casts them: from the camera into the scene, then from the points those rays hit (float32 origins that lie
    ON a triangle, where a box test has no slack) in random, axis-aligned and grazing directions, long and short."""
    rng = np.random.default_rng(seed)
    cam = np.linalg.inv(np.asarray(s.world_to_view, np.float64))[:3, 3].astype(np.float32)
    def unit(k):
        d = rng.standard_normal((k, 3)); return (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    first = np.zeros((n, 8), np.float32); first[:, 0:3] = cam; first[:, 4:7] = unit(n); first[:, 7] = 1000.0
    h = o.intersect_many(first)
    hit = h[:, 0] > 0
    origin = (first[hit, 0:3] + h[hit, 1:2] * first[hit, 4:7]).astype(np.float32)
    m = len(origin)
    d = unit(m)
    axis = np.eye(3, dtype=np.float32)[rng.integers(0, 3, m)] * rng.choice(np.float32([-1, 1]), m)[:, None]
    kind = rng.integers(0, 4, m)
    d[kind == 1] = axis[kind == 1]                                                   # along a world axis: two zero direction components ...
    # ... from an origin moved off the surface: a ray that runs exactly IN the plane of an axis-aligned wall, along a triangle edge, is in or
    # out of a box by the sign of a zero, which neither DXR nor this pair of traversals defines
    origin[kind == 1] += (rng.random((int((kind == 1).sum()), 3)).astype(np.float32) - 0.5) * np.float32(2e-3)
    d[kind == 2] = (d[kind == 2] * np.float32([1, 0.02, 1])); d[kind == 2] /= np.linalg.norm(d[kind == 2], axis=1, keepdims=True)   # grazing the floors and ceilings
    second = np.zeros((m, 8), np.float32); second[:, 0:3] = origin; second[:, 4:7] = d
    second[:, 7] = np.where(rng.random(m) < 0.25, rng.random(m) * 0.5, 1000.0).astype(np.float32)
    return first, second
