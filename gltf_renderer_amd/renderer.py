"""ctypes binding of libmipt.so (include/mipt.h).  No CPU fallback: if the HIP library is missing or
its symbols do not match the header, importing the Renderer fails loudly.

PyTorch is used only as plumbing: the caller-owned output image is a CUDA(HIP) tensor and the
library enqueues on torch's current stream, so torch.distributed (RCCL) can reduce the image.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmipt.so")

# every symbol include/mipt.h declares
EXPORTS = ["pt_abi_version", "pt_create", "pt_destroy", "pt_last_error", "pt_buffer_create", "pt_buffer_update", "pt_buffer_read",
           "pt_texture_create", "pt_sampler_create", "pt_scene_set_materials", "pt_scene_set_lights", "pt_scene_set_instances",
           "pt_env_create", "pt_env_read", "pt_build_accel", "pt_skin_run", "pt_trace", "pt_set_bounce_limit", "pt_set_samples_per_trace", "pt_set_null_shadow_culling", "pt_enable_counters",
           "pt_set_kernel_mode",
           "pt_get_stats", "pt_reset_stats", "pt_readback", "pt_tonemap",
           # ABI 2
           "pt_buffer_destroy", "pt_texture_destroy", "pt_env_destroy", "pt_accel_request_rebuild", "pt_set_accel_builder", "pt_enable_stage_timing",
           "pt_exchange_unique_id", "pt_exchange_probe", "pt_exchange_create", "pt_exchange_create_loopback", "pt_exchange_frame", "pt_exchange_destroy",
           "pt_tiles_packed_bytes", "pt_tiles_pack", "pt_tiles_unpack"]


class MiptError(RuntimeError):
    pass


_LIB = None


def load_library():
    global _LIB
    if _LIB is not None:
        return _LIB
    # MIPT_LIBRARY: a tuning build of the same library (tools/build_variant.sh) for A/B runs; never another implementation
    lib_path = os.environ.get("MIPT_LIBRARY") or LIB_PATH
    if not os.path.exists(lib_path):
        raise MiptError("libmipt.so not found at %s: build it with `make -C gltf_renderer_amd/csrc` "
                        "(or __graft_entry__.build()); there is no CPU fallback" % lib_path)
    # One HIP runtime per process: PyTorch ships its own libamdhip64, and whichever copy is mapped first serves both.  Import
    # torch BEFORE libmipt.so so that it is torch's copy (loading /opt/rocm's first makes later device queries fail).
    import torch  # noqa: F401
    L = C.CDLL(lib_path)
    missing = [s for s in EXPORTS if not hasattr(L, s)]
    if missing:
        raise MiptError("libmipt.so lacks symbols declared in include/mipt.h: %s" % missing)
    if L.pt_abi_version() != 2:
        raise MiptError("libmipt.so ABI version mismatch")
    vp, ci = C.c_void_p, C.c_int
    L.pt_create.argtypes = [ci, vp, vp, C.POINTER(vp)]
    L.pt_destroy.argtypes = [vp]
    L.pt_destroy.restype = None
    L.pt_last_error.argtypes = [vp]
    L.pt_last_error.restype = C.c_char_p
    L.pt_buffer_create.argtypes = [vp, vp, C.c_size_t, ci, C.POINTER(ci)]
    L.pt_buffer_update.argtypes = [vp, ci, vp, C.c_size_t]
    L.pt_buffer_read.argtypes = [vp, ci, vp, C.c_size_t]
    L.pt_texture_create.argtypes = [vp, vp, ci, ci, ci, C.POINTER(ci)]
    L.pt_sampler_create.argtypes = [vp, vp, C.POINTER(ci)]
    L.pt_scene_set_materials.argtypes = [vp, vp, ci]
    L.pt_scene_set_lights.argtypes = [vp, vp, ci]
    L.pt_scene_set_instances.argtypes = [vp, vp, ci]
    L.pt_env_create.argtypes = [vp, vp, ci, ci, C.POINTER(ci)]
    L.pt_env_read.argtypes = [vp, ci, C.POINTER(ci), vp, vp]
    L.pt_build_accel.argtypes = [vp]
    L.pt_skin_run.argtypes = [vp, vp, vp, ci]
    L.pt_trace.argtypes = [vp, vp, vp]
    L.pt_set_bounce_limit.argtypes = [vp, ci]
    L.pt_set_samples_per_trace.argtypes = [vp, ci]
    L.pt_set_null_shadow_culling.argtypes = [vp, ci]
    L.pt_enable_counters.argtypes = [vp, ci]
    L.pt_set_kernel_mode.argtypes = [vp, ci, ci]
    L.pt_get_stats.argtypes = [vp, vp]
    L.pt_reset_stats.argtypes = [vp]
    L.pt_readback.argtypes = [vp, vp, C.c_uint32, C.c_uint32, vp]
    L.pt_tonemap.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, vp, vp]
    L.pt_buffer_destroy.argtypes = [vp, ci]
    L.pt_texture_destroy.argtypes = [vp, ci]
    L.pt_env_destroy.argtypes = [vp, ci]
    L.pt_accel_request_rebuild.argtypes = [vp]
    L.pt_set_accel_builder.argtypes = [vp, ci]
    L.pt_enable_stage_timing.argtypes = [vp, ci]
    L.pt_exchange_unique_id.argtypes = [vp]
    L.pt_exchange_create.argtypes = [vp, ci, ci, vp]
    L.pt_exchange_probe.argtypes = []
    L.pt_exchange_create_loopback.argtypes = [vp, ci, ci, C.c_uint64]
    L.pt_exchange_frame.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, ci, ci]
    L.pt_exchange_destroy.argtypes = [vp]
    L.pt_tiles_packed_bytes.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    L.pt_tiles_packed_bytes.restype = C.c_size_t
    L.pt_tiles_pack.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp]
    L.pt_tiles_unpack.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp]
    _LIB = L
    return L


def sheen_lut():
    return np.fromfile(os.path.join(_HERE, "data", "sheen_e_16x16.f32"), dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Renderer:
    """Host mirror of the reference's hot-path objects behind one context:
    Pathtracer::{Init, PathtraceScene, Shutdown} (Source/Pathtracer.h:104-106),
    GpuSkin::{Create, Run} (Source/GpuSkin.h:17-19), EnvironmentMap::CreateEnvironmentMap."""

    def __init__(self, device=0, stream=None):
        import torch
        if not torch.cuda.is_available():
            raise MiptError("no HIP device visible: the path tracer runs on MI355X only (no CPU fallback)")
        self.torch = torch
        self.L = load_library()
        self.device = device
        torch.cuda.set_device(device)
        self.stream = torch.cuda.current_stream(device).cuda_stream if stream is None else stream
        lut = sheen_lut()
        h = C.c_void_p()
        rc = self.L.pt_create(device, C.c_void_p(self.stream), _p(lut), C.byref(h))
        if rc != 0:
            raise MiptError("pt_create failed: %d" % rc)
        self.h = h

    def _check(self, rc):
        if rc != 0:
            raise MiptError("%d: %s" % (rc, self.L.pt_last_error(self.h).decode()))

    def close(self):
        if getattr(self, "h", None):
            self.L.pt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- resources
    def buffer_create(self, data, fmt, nbytes=None):
        out = C.c_int()
        if data is None:
            self._check(self.L.pt_buffer_create(self.h, None, nbytes, fmt, C.byref(out)))
        else:
            a = np.ascontiguousarray(data)
            self._check(self.L.pt_buffer_create(self.h, _p(a), a.nbytes, fmt, C.byref(out)))
        return out.value

    def buffer_update(self, handle, data):
        a = np.ascontiguousarray(data)
        self._check(self.L.pt_buffer_update(self.h, handle, _p(a), a.nbytes))

    def buffer_read(self, handle, dtype, count):
        out = np.zeros(count, dtype)
        self._check(self.L.pt_buffer_read(self.h, handle, _p(out), out.nbytes))
        return out

    def buffer_destroy(self, handle):
        self._check(self.L.pt_buffer_destroy(self.h, handle))

    def texture_destroy(self, handle):
        self._check(self.L.pt_texture_destroy(self.h, handle))

    def env_destroy(self, env):
        self._check(self.L.pt_env_destroy(self.h, env))

    def texture_create(self, rgba8, srgb):
        a = np.ascontiguousarray(rgba8, dtype=np.uint8)
        h, w = a.shape[:2]
        out = C.c_int()
        self._check(self.L.pt_texture_create(self.h, _p(a), w, h, int(bool(srgb)), C.byref(out)))
        return out.value

    def sampler_create(self, address_u, address_v, min_filter, mag_filter):
        d = abi.PtSamplerDesc(address_u, address_v, min_filter, mag_filter)
        out = C.c_int()
        self._check(self.L.pt_sampler_create(self.h, C.byref(d), C.byref(out)))
        return out.value

    def set_materials(self, materials):
        arr = (abi.PtMaterial * len(materials))(*materials)
        self._check(self.L.pt_scene_set_materials(self.h, C.byref(arr), len(materials)))

    def set_lights(self, lights):
        if len(lights) == 0:
            self._check(self.L.pt_scene_set_lights(self.h, None, 0))
            return
        arr = (abi.PtLight * len(lights))(*lights)
        self._check(self.L.pt_scene_set_lights(self.h, C.byref(arr), len(lights)))

    def set_instances(self, instances):
        arr = (abi.PtInstanceDesc * len(instances))(*instances)
        self._check(self.L.pt_scene_set_instances(self.h, C.byref(arr), len(instances)))

    def env_create(self, equirect_rgb32f):
        a = np.ascontiguousarray(equirect_rgb32f, dtype=np.float32)
        h, w = a.shape[:2]
        out = C.c_int()
        self._check(self.L.pt_env_create(self.h, _p(a), w, h, C.byref(out)))
        return out.value

    def env_read(self, env):
        n = C.c_int()
        self._check(self.L.pt_env_read(self.h, env, C.byref(n), None, None))
        cube = np.zeros((6, n.value, n.value, 4), np.uint16)
        pyr = np.zeros(sum((1024 >> i) ** 2 for i in range(11)), np.float32)
        self._check(self.L.pt_env_read(self.h, env, C.byref(n), _p(cube), _p(pyr)))
        return n.value, cube, pyr

    def set_bounce_limit(self, limit):
        self._check(self.L.pt_set_bounce_limit(self.h, limit))

    def set_samples_per_trace(self, samples):
        """Sample batch: one trace() then stands for `samples` consecutive frames (bit-identical to issuing them one by one)."""
        self._check(self.L.pt_set_samples_per_trace(self.h, int(samples)))

    def set_null_shadow_culling(self, on):
        """Skip shadow rays whose contribution is exactly zero (same image, fewer rays than the reference traces)."""
        self._check(self.L.pt_set_null_shadow_culling(self.h, int(bool(on))))

    def enable_counters(self, on):
        self._check(self.L.pt_enable_counters(self.h, int(on)))

    def set_kernel_mode(self, mode, stage_blocks=0):
        """mode: abi.MODE_WAVEFRONT (default) or abi.MODE_MEGAKERNEL."""
        self._check(self.L.pt_set_kernel_mode(self.h, mode, stage_blocks))

    def build_accel(self):
        """Full build, refit or nothing, whichever the changes since the last call ask for (include/mipt.h pt_build_accel)."""
        self._check(self.L.pt_build_accel(self.h))

    def request_rebuild(self):
        self._check(self.L.pt_accel_request_rebuild(self.h))

    def set_accel_builder(self, builder):
        """abi.BUILDER_LBVH (radix tree), abi.BUILDER_PLOC (clustering by surface area: better tree, slower build) or
        abi.BUILDER_PLOC_REINSERT (the default: the PLOC tree improved by parallel reinsertion passes)."""
        self._check(self.L.pt_set_accel_builder(self.h, int(builder)))

    def enable_stage_timing(self, on):
        self._check(self.L.pt_enable_stage_timing(self.h, int(bool(on))))

    # ---- multi-GPU exchange (RCCL inside libmipt.so; the unique id travels by whatever transport the host has)
    @staticmethod
    def exchange_probe():
        """True if RCCL can be loaded in this process (no GPU call): ranks agree on this before any enters the collective create."""
        return load_library().pt_exchange_probe() == 0

    def exchange_unique_id(self):
        buf = (C.c_ubyte * abi.EXCHANGE_ID_BYTES)()
        rc = self.L.pt_exchange_unique_id(buf)
        if rc != 0:
            raise MiptError("pt_exchange_unique_id failed: %d (RCCL not loadable?)" % rc)
        return bytes(buf)

    def exchange_create(self, rank, world, unique_id=None):
        buf = None if unique_id is None else (C.c_ubyte * abi.EXCHANGE_ID_BYTES).from_buffer_copy(unique_id)
        self._check(self.L.pt_exchange_create(self.h, rank, world, buf))

    def exchange_create_loopback(self, rank, world, group):
        """N contexts of this process exchange by device-to-device copies (no RCCL); call exchange_frame on the root last."""
        self._check(self.L.pt_exchange_create_loopback(self.h, rank, world, C.c_uint64(group)))

    def exchange_frame(self, local, frame=None, mode=abi.EXCHANGE_GATHER, dst=0):
        """local: this rank's accumulation image (H, W, 4) CUDA tensor, only read.  frame: where the root assembles the frame
        (None = in place over the other ranks' tiles of `local`).  Asynchronous on the context's stream."""
        h, w = local.shape[:2]
        self._check(self.L.pt_exchange_frame(self.h, C.c_void_p(local.data_ptr()), C.c_void_p(frame.data_ptr()) if frame is not None else None, w, h, mode, dst))

    def exchange_destroy(self):
        self._check(self.L.pt_exchange_destroy(self.h))

    def tiles_packed_bytes(self, width, height, rank, world):
        return int(self.L.pt_tiles_packed_bytes(width, height, rank, world))

    def tiles_pack(self, image, rank, world):
        h, w = image.shape[:2]
        n = self.tiles_packed_bytes(w, h, rank, world) // 16
        packed = self.torch.empty((max(n, 1), 4), dtype=self.torch.float32, device=image.device)
        self._check(self.L.pt_tiles_pack(self.h, C.c_void_p(image.data_ptr()), w, h, rank, world, C.c_void_p(packed.data_ptr())))
        return packed[:n]

    def tiles_unpack(self, packed, image, rank, world):
        h, w = image.shape[:2]
        assert packed.numel() * 4 >= self.tiles_packed_bytes(w, h, rank, world)
        self._check(self.L.pt_tiles_unpack(self.h, C.c_void_p(packed.data_ptr()), w, h, rank, world, C.c_void_p(image.data_ptr())))

    def skin_run(self, params, bones):
        if bones is None or len(bones) == 0:
            self._check(self.L.pt_skin_run(self.h, C.byref(params), None, 0))
        else:
            arr = (abi.PtBone * len(bones))(*bones)
            self._check(self.L.pt_skin_run(self.h, C.byref(params), C.byref(arr), len(bones)))

    # ---- rendering
    def create_output(self, width, height):
        return self.torch.zeros((height, width, 4), dtype=self.torch.float32, device="cuda:%d" % self.device)

    def trace(self, settings, params, output):
        """output: torch float32 CUDA tensor (H, W, 4), the caller-owned accumulation target."""
        assert output.is_cuda and output.is_contiguous() and output.dtype == self.torch.float32
        assert tuple(output.shape) == (params.height, params.width, 4)
        params.output = output.data_ptr()
        self._check(self.L.pt_trace(self.h, C.byref(settings), C.byref(params)))

    def reset_stats(self):
        self._check(self.L.pt_reset_stats(self.h))

    def stats(self):
        s = abi.PtStats()
        self._check(self.L.pt_get_stats(self.h, C.byref(s)))
        return s

    def readback(self, output):
        h, w = output.shape[:2]
        out = np.zeros((h, w, 4), np.float32)
        self._check(self.L.pt_readback(self.h, C.c_void_p(output.data_ptr()), w, h, _p(out)))
        return out

    def tonemap(self, output, config=None, want_rgba8=False):
        cfg = config or abi.PtTonemapConfig.default()
        h, w = output.shape[:2]
        rgb = np.zeros((h, w, 3), np.float32)
        q = np.zeros((h, w, 4), np.uint8) if want_rgba8 else None
        self._check(self.L.pt_tonemap(self.h, C.byref(cfg), C.c_void_p(output.data_ptr()), w, h, _p(rgb), _p(q) if want_rgba8 else None))
        return (rgb, q) if want_rgba8 else rgb
