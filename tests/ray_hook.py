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


MATH_OPS = {"atan2": 0, "pow": 1, "exp": 2, "log2": 3, "exp2": 4, "sin": 5, "cos": 6, "div": 7, "pow5": 8}


def math_inputs(seed=3, n=400_000):
    """Arguments for the co-defined math routines: wide random ranges plus every special value."""
    rng = np.random.default_rng(seed)
    special = np.float32([0.0, -0.0, 1.0, -1.0, 2.0, 0.5, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1.1754944e-38, 3.4028235e38, -3.4028235e38, 1e-30, 88.75, -86.5, 128.0, -125.0, 127.5, -124.5])
    wide = lambda: (rng.standard_normal(n) * 10.0 ** rng.uniform(-8, 8, n)).astype(np.float32)
    out = {}
    sa, sb = np.meshgrid(special, special)
    out["atan2"] = (np.concatenate([wide(), sa.ravel(), rng.standard_normal(n).astype(np.float32)]), np.concatenate([wide(), sb.ravel(), rng.standard_normal(n).astype(np.float32)]))
    out["div"] = out["atan2"]
    base = np.concatenate([rng.uniform(0, 1, n), 10.0 ** rng.uniform(-30, 30, n), sa.ravel()]).astype(np.float32)
    expo = np.concatenate([rng.uniform(0, 60, n), rng.uniform(-4, 4, n), sb.ravel()]).astype(np.float32)
    out["pow"] = (base, expo)
    e = np.concatenate([rng.uniform(-100, 100, n), rng.uniform(-6, 1, n), special]).astype(np.float32)
    out["exp"] = (e, e)
    p = np.concatenate([rng.uniform(-140, 140, n), special]).astype(np.float32)
    out["exp2"] = (p, p)
    l = np.concatenate([np.abs(wide()), rng.uniform(0.5, 2, n).astype(np.float32), special]).astype(np.float32)
    out["log2"] = (l, l)
    t = np.concatenate([rng.uniform(-7, 7, n), wide(), special]).astype(np.float32)
    out["sin"] = (t, t); out["cos"] = (t, t)
    f = np.concatenate([rng.uniform(-0.5, 1.5, n), special]).astype(np.float32)
    out["pow5"] = (f, f)
    return out


def oracle_math(lib, op, a, b):
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32); out = np.zeros_like(a)
    lib.orc_math.restype = None
    lib.orc_math(C.c_int(MATH_OPS[op]), a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_int(len(a)))
    return out


def gpu_math(lib, op, a, b):
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32); out = np.zeros_like(a)
    lib.pt_debug_math.restype = C.c_int
    rc = lib.pt_debug_math(C.c_int(MATH_OPS[op]), a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_uint32(len(a)))
    assert rc == 0, rc
    return out
