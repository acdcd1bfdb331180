"""ctypes wrapper of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, and by nothing
under gltf_renderer_amd/.  It reuses gltf_renderer_amd.abi only for the plain-data layouts of the
C-ABI contract (that is data description, not product code).
"""
import ctypes as C
import os
import subprocess

import numpy as np

from gltf_renderer_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_void_p]
        for name in ("orc_destroy", "orc_build_accel"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = None
        L.orc_build_accel_mt.argtypes = [C.c_void_p, C.c_int]
        L.orc_build_accel_mt.restype = None
        L.orc_buffer_create.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.orc_buffer_update.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.orc_buffer_read.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.orc_texture_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_sampler_create.argtypes = [C.c_void_p, C.c_void_p]
        for name in ("orc_set_materials", "orc_set_lights", "orc_set_instances"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p, C.c_int]
            getattr(L, name).restype = None
        L.orc_env_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.orc_env_create_raw.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_env_cube_size.argtypes = [C.c_void_p, C.c_int]
        L.orc_env_read.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_env_read.restype = None
        L.orc_set_bounce_limit.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_brute_force.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_window.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_set_window.restype = None
        L.orc_set_ray_log.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_set_ray_log.restype = None
        L.orc_read_ray_log.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_read_ray_log.restype = C.c_int
        L.orc_skin_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_trace.restype = None
        L.orc_get_counters.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_get_timing.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_bvh_info.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_tonemap.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_sheen_e.restype = C.c_float
        L.orc_sheen_e.argtypes = [C.c_void_p, C.c_float, C.c_float]
        L.orc_importance_map_pdf.restype = C.c_float
        L.orc_importance_map_pdf.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_half_to_float.restype = C.c_float
        L.orc_half_to_float.argtypes = [C.c_uint16]
        L.orc_float_to_half.restype = C.c_uint16
        L.orc_float_to_half.argtypes = [C.c_float]
        L.orc_encode_tangent_space_host.restype = C.c_uint32
        L.orc_encode_tangent_space_shader.restype = C.c_uint32
        L.orc_encode_normal_host.restype = C.c_uint32
        L.orc_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_uint32, C.c_void_p]
        L.orc_sample_cube.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_void_p]
        L.orc_evaluate_bsdf.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_sample_bsdf.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_sample_texture.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_sample_importance_map.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_decode_tangent_space.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_random.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
        L.orc_cubemap_to_direction.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


def sheen_lut():
    p = os.path.join(_HERE, "..", "tests", "golden", "sheen_e_16x16.npy")
    return np.ascontiguousarray(np.load(p).astype(np.float32))


def _f(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# ---- vector helpers for the unit-level entry points ---------------------------------------------
def call_vec(name, inp, n_out, *pre):
    L = lib()
    a = _f(inp)
    out = np.zeros(n_out, np.float32)
    getattr(L, name)(*pre, _p(a), _p(out))
    return out


class Oracle:
    """Same method surface as gltf_renderer_amd.Renderer so a scene uploads to either."""

    def __init__(self, lut=None):
        self.L = lib()
        self.lut = sheen_lut() if lut is None else _f(lut)
        self.h = C.c_void_p(self.L.orc_create(_p(self.lut)))
        self._keep = []

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def buffer_create(self, data, fmt, nbytes=None):
        if data is None:
            return self.L.orc_buffer_create(self.h, None, nbytes, fmt)
        a = np.ascontiguousarray(data)
        return self.L.orc_buffer_create(self.h, _p(a), a.nbytes, fmt)

    def buffer_update(self, handle, data):
        a = np.ascontiguousarray(data)
        self.L.orc_buffer_update(self.h, handle, _p(a), a.nbytes)

    def buffer_read(self, handle, dtype, count):
        out = np.zeros(count, dtype)
        self.L.orc_buffer_read(self.h, handle, _p(out), out.nbytes)
        return out

    def texture_create(self, rgba8, srgb):
        a = np.ascontiguousarray(rgba8, dtype=np.uint8)
        h, w = a.shape[:2]
        return self.L.orc_texture_create(self.h, _p(a), w, h, int(bool(srgb)))

    def sampler_create(self, address_u, address_v, min_filter, mag_filter):
        d = np.array([address_u, address_v, min_filter, mag_filter], np.int32)
        return self.L.orc_sampler_create(self.h, _p(d))

    def set_materials(self, materials):
        arr = (abi.PtMaterial * len(materials))(*materials)
        self.L.orc_set_materials(self.h, C.byref(arr), len(materials))

    def set_lights(self, lights):
        if len(lights) == 0:
            self.L.orc_set_lights(self.h, None, 0)
            return
        arr = (abi.PtLight * len(lights))(*lights)
        self.L.orc_set_lights(self.h, C.byref(arr), len(lights))

    def set_instances(self, instances):
        arr = (abi.PtInstanceDesc * len(instances))(*instances)
        self.L.orc_set_instances(self.h, C.byref(arr), len(instances))

    def env_create(self, equirect_rgb32f):
        a = _f(equirect_rgb32f)
        h, w = a.shape[:2]
        return self.L.orc_env_create(self.h, _p(a), w, h)

    def env_create_raw(self, cube_size, cube_rgba16f, pyramid):
        c = np.ascontiguousarray(cube_rgba16f, dtype=np.uint16)
        p = _f(pyramid)
        return self.L.orc_env_create_raw(self.h, cube_size, _p(c), _p(p))

    def env_read(self, env):
        n = self.L.orc_env_cube_size(self.h, env)
        cube = np.zeros((6, n, n, 4), np.uint16)
        pyr = np.zeros(sum((1024 >> i) ** 2 for i in range(11)), np.float32)
        self.L.orc_env_read(self.h, env, _p(cube), _p(pyr))
        return n, cube, pyr

    def set_bounce_limit(self, limit):
        self.L.orc_set_bounce_limit(self.h, limit)

    def set_brute_force(self, on):
        self.L.orc_set_brute_force(self.h, int(on))

    def set_window(self, x0=0, y0=0, x1=0, y1=0):
        """Diagnostics: trace() touches only the pixels x0 <= x < x1, y0 <= y < y1 (no arguments: the whole frame again)."""
        self.L.orc_set_window(self.h, x0, y0, x1, y1)

    def ray_log(self, x, y):
        """Diagnostics: start recording every ray of pixel (x, y) (x < 0: stop)."""
        self.L.orc_set_ray_log(self.h, x, y)

    def read_ray_log(self):
        """[n, 16] float32: origin, tmin, direction, tmax, mode (0 closest, 1 shadow), committed, t, instance, primitive, transmission, D3D12 ray flags, 0."""
        n = self.L.orc_read_ray_log(self.h, None, 0)
        out = np.zeros((n, 16), np.float32)
        if n: self.L.orc_read_ray_log(self.h, out.ctypes.data, n)
        return out

    def build_accel(self, nthreads=1):
        """CPU LBVH of the scene; nthreads > 1 builds the SAME tree (node for node) on that many cores."""
        if nthreads and nthreads > 1:
            self.L.orc_build_accel_mt(self.h, int(nthreads))
        else:
            self.L.orc_build_accel(self.h)

    def skin_run(self, params, bones):
        if bones is None or len(bones) == 0:
            self.L.orc_skin_run(self.h, C.byref(params), None, 0)
        else:
            arr = (abi.PtBone * len(bones))(*bones)
            self.L.orc_skin_run(self.h, C.byref(params), C.byref(arr), len(bones))

    def trace(self, settings, params, output, nthreads=None):
        """output: float32 array (H, W, 4), modified in place (the accumulation target)."""
        assert output.dtype == np.float32 and output.flags["C_CONTIGUOUS"]
        params.output = output.ctypes.data
        if nthreads is None:
            nthreads = os.cpu_count() or 1
        self.L.orc_trace(self.h, C.byref(settings), C.byref(params), nthreads)

    def counters(self, reset=True):
        out = np.zeros(8, np.uint64)
        self.L.orc_get_counters(self.h, _p(out), int(reset))
        keys = ["primary", "bounce", "shadow", "nodes", "tris", "closest_hits", "texture_taps", "accumulated_frames"]
        d = dict(zip(keys, (int(x) for x in out)))
        d["rays"] = d["primary"] + d["bounce"] + d["shadow"]
        return d

    def timing(self):
        a, t = C.c_double(), C.c_double()
        self.L.orc_get_timing(self.h, C.byref(a), C.byref(t))
        return a.value, t.value

    def bvh_info(self):
        n, t = C.c_uint32(), C.c_uint32()
        self.L.orc_bvh_info(self.h, C.byref(n), C.byref(t))
        return n.value, t.value

    def intersect_many(self, rays, ray_flags=0, mode=0, nthreads=None):
        """rays [n, 8] (origin, tmin, direction, tmax) -> [n, 8] (committed, t, u, v, instance, primitive, front, transmission)."""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        out = np.zeros((len(rays), 8), np.float32)
        self.L.orc_intersect_many.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_int, C.c_void_p, C.c_int]
        self.L.orc_intersect_many.restype = None
        self.L.orc_intersect_many(self.h, rays.ctypes.data, len(rays), ray_flags, mode, out.ctypes.data, nthreads or os.cpu_count() or 1)
        return out

    def intersect(self, origin, direction, tmin=0.0, tmax=1e30, ray_flags=0):
        o, d = _f(origin), _f(direction)
        out = np.zeros(7, np.float32)
        self.L.orc_intersect(self.h, _p(o), _p(d), tmin, tmax, ray_flags, _p(out))
        return out


def tonemap(rgba, config=None, want_rgba8=False):
    L = lib()
    cfg = config or abi.PtTonemapConfig.default()
    a = _f(rgba)
    h, w = a.shape[:2]
    rgb = np.zeros((h, w, 3), np.float32)
    q = np.zeros((h, w, 4), np.uint8) if want_rgba8 else None
    L.orc_tonemap(C.byref(cfg), _p(a), w, h, _p(rgb), _p(q) if want_rgba8 else None)
    return (rgb, q) if want_rgba8 else rgb
