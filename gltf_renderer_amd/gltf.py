"""ctypes binding of the scene side of libmipt.so (include/mipt_scene.h): class Gltf of the reference
(Source/Gltf.h:16-232) as loaded by the C++ loader, its animation player, the per-frame host walk
(Renderer.cpp:399-500, Pathtracer.cpp:185-257) and the image-file readers.  No work happens in Python:
every method is one C-ABI call plus numpy views of the returned arrays."""
import ctypes as C

import numpy as np

from . import abi
from .renderer import MiptError, load_library

SCENE_EXPORTS = ["gs_load_file", "gs_free", "gs_last_error", "gs_get_counts", "gs_get_primitive", "gs_get_morph_target", "gs_get_material",
                 "gs_get_texture", "gs_get_sampler", "gs_get_camera", "gs_get_node", "gs_get_node_weights", "gs_get_scene_nodes", "gs_get_skin", "gs_get_animation",
                 "gs_get_channel", "gs_sample_channel", "gs_apply_rest_transforms", "gs_animate", "gs_calculate_global_transforms", "gs_player_tick",
                 "gs_gather_lights", "gs_gather_bones", "gs_upload", "gs_unload", "gs_frame", "img_load_rgba8", "img_decode_rgba8", "img_load_rgb32f",
                 "img_decode_rgb32f", "img_free", "img_write_png", "img_write_pfm", "img_write_exr"]


class GsCounts(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("meshes", "primitives", "materials", "nodes", "scenes", "skins", "animations", "lights", "textures", "samplers",
                                       "cameras", "dynamic_meshes")]


class GsPrimitiveInfo(C.Structure):
    _fields_ = [("mesh", C.c_int), ("index_in_mesh", C.c_int), ("flags", C.c_int), ("topology", C.c_int), ("num_vertices", C.c_int), ("num_indices", C.c_int),
                ("index_format", C.c_int), ("material_id", C.c_int), ("num_targets", C.c_int), ("index", C.c_void_p), ("position", C.c_void_p),
                ("tangent_space", C.c_void_p), ("texcoord", C.c_void_p * 2), ("color", C.c_void_p), ("joint_weight", C.c_void_p)]


class GsCameraInfo(C.Structure):
    _fields_ = [("type", C.c_int), ("aspect_ratio", C.c_float), ("y_fov", C.c_float), ("x_mag", C.c_float), ("y_mag", C.c_float), ("z_near", C.c_float), ("z_far", C.c_float),
                ("upstream_type_matches", C.c_int), ("view_to_clip", C.c_float * 16)]


class GsNodeInfo(C.Structure):
    _fields_ = [("child", C.c_int), ("sibling", C.c_int), ("mesh", C.c_int), ("skin", C.c_int), ("dynamic_mesh", C.c_int), ("camera", C.c_int), ("light", C.c_int),
                ("rest_translation", C.c_float * 3), ("rest_rotation", C.c_float * 4), ("rest_scale", C.c_float * 3),
                ("local_translation", C.c_float * 3), ("local_rotation", C.c_float * 4), ("local_scale", C.c_float * 3),
                ("global_transform", C.c_float * 16), ("num_current_weights", C.c_int)]


class GsChannelInfo(C.Structure):
    _fields_ = [("node", C.c_int), ("path", C.c_int), ("interpolation", C.c_int), ("format", C.c_int), ("width", C.c_int), ("num_times", C.c_int),
                ("num_transform_bytes", C.c_int), ("times", C.c_void_p), ("transforms", C.c_void_p)]


class GsPlayer(C.Structure):
    _fields_ = [("animation", C.c_int), ("playhead", C.c_float), ("playing", C.c_int), ("loop", C.c_int)]


_READY = False


def _lib():
    global _READY
    L = load_library()
    if not _READY:
        missing = [s for s in SCENE_EXPORTS if not hasattr(L, s)]
        if missing:
            raise MiptError("libmipt.so lacks symbols declared in include/mipt_scene.h: %s" % missing)
        L.gs_last_error.restype = C.c_char_p
        L.gs_free.restype = None
        L.img_free.restype = None
        L.img_free.argtypes = [C.c_void_p]
        L.gs_free.argtypes = [C.c_void_p]
        L.gs_load_file.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        for name in ("gs_get_counts", "gs_get_primitive", "gs_get_material", "gs_get_sampler", "gs_get_camera", "gs_get_node", "gs_get_channel"):
            getattr(L, name).argtypes = None
        L.gs_sample_channel.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]
        L.gs_animate.argtypes = [C.c_void_p, C.c_int, C.c_float]
        L.gs_player_tick.argtypes = [C.c_void_p, C.c_void_p, C.c_float]
        L.img_decode_rgba8.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        L.img_decode_rgb32f.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _READY = True
    return L


def _err(L, rc):
    raise MiptError("%d: %s" % (rc, L.gs_last_error().decode(errors="replace")))


def _view(ptr, dtype, count):
    if not ptr or count == 0:
        return None
    buf = (C.c_char * (np.dtype(dtype).itemsize * count)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=count).copy()


class GltfScene:
    """class Gltf (Source/Gltf.h): meshes / materials / nodes / skins / animations / lights / textures of one file."""

    def __init__(self, path):
        self.L = _lib()
        h = C.c_void_p()
        rc = self.L.gs_load_file(str(path).encode(), C.byref(h))
        if rc != 0:
            _err(self.L, rc)
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.gs_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc < 0:
            _err(self.L, rc)
        return rc

    def counts(self):
        c = GsCounts()
        self._ck(self.L.gs_get_counts(self.h, C.byref(c)))
        return c

    def primitive(self, flat_index):
        """dict of numpy arrays in the exact stream formats of Mesh.cpp:124-132."""
        p = GsPrimitiveInfo()
        self._ck(self.L.gs_get_primitive(self.h, C.c_int(flat_index), C.byref(p)))
        nv, ni = p.num_vertices, p.num_indices
        out = {"mesh": p.mesh, "index_in_mesh": p.index_in_mesh, "flags": p.flags, "topology": p.topology, "num_vertices": nv, "num_indices": ni,
               "index_format": p.index_format, "material_id": p.material_id, "num_targets": p.num_targets}
        out["index"] = _view(p.index, np.uint16 if p.index_format == abi.FORMAT_R16_UINT else np.uint32, ni)
        pos = _view(p.position, np.float32, nv * 3)
        out["position"] = None if pos is None else pos.reshape(-1, 3)
        out["tangent_space"] = _view(p.tangent_space, np.uint32, nv)
        for k in range(2):
            t = _view(p.texcoord[k], np.float32, nv * 2)
            out["texcoord%d" % k] = None if t is None else t.reshape(-1, 2)
        c = _view(p.color, np.uint16, nv * 4)
        out["color"] = None if c is None else c.reshape(-1, 4)
        jw = _view(p.joint_weight, np.uint16, nv * 8)
        out["joint_weight"] = None if jw is None else jw.reshape(-1, 8)
        return out

    def morph_target(self, flat_index, target, num_vertices):
        flags, pos, ts = C.c_int(), C.c_void_p(), C.c_void_p()
        self._ck(self.L.gs_get_morph_target(self.h, C.c_int(flat_index), C.c_int(target), C.byref(flags), C.byref(pos), C.byref(ts)))
        p = _view(pos.value, np.float32, num_vertices * 3)
        return flags.value, (None if p is None else p.reshape(-1, 3)), _view(ts.value, np.uint32, num_vertices)

    def material(self, i):
        m = abi.PtMaterial()
        self._ck(self.L.gs_get_material(self.h, C.c_int(i), C.byref(m)))
        return m

    def texture(self, i):
        w, h, srgb, loaded, ptr = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_void_p()
        self._ck(self.L.gs_get_texture(self.h, C.c_int(i), C.byref(w), C.byref(h), C.byref(srgb), C.byref(loaded), C.byref(ptr)))
        px = _view(ptr.value, np.uint8, w.value * h.value * 4) if loaded.value else None
        return {"width": w.value, "height": h.value, "srgb": bool(srgb.value), "loaded": bool(loaded.value),
                "rgba": None if px is None else px.reshape(h.value, w.value, 4)}

    def sampler(self, i):
        d = abi.PtSamplerDesc()
        self._ck(self.L.gs_get_sampler(self.h, C.c_int(i), C.byref(d)))
        return d

    def camera(self, i):
        """Gltf::LoadCameras: the file's camera `i` and Camera::GetViewToClip's matrix for it."""
        c = GsCameraInfo()
        self._ck(self.L.gs_get_camera(self.h, C.c_int(i), C.byref(c)))
        return c

    def node(self, i):
        n = GsNodeInfo()
        self._ck(self.L.gs_get_node(self.h, C.c_int(i), C.byref(n)))
        return n

    def node_weights(self, i):
        out = np.zeros(64, np.float32)
        n = self._ck(self.L.gs_get_node_weights(self.h, C.c_int(i), out.ctypes.data_as(C.c_void_p), C.c_int(64)))
        return out[:n].copy()

    def scene_nodes(self, scene=0):
        out = np.zeros(4096, np.int32)
        n = self._ck(self.L.gs_get_scene_nodes(self.h, C.c_int(scene), out.ctypes.data_as(C.c_void_p), C.c_int(4096)))
        return out[:n].tolist()

    def skin(self, i):
        nj, joints, ibp = C.c_int(), C.c_void_p(), C.c_void_p()
        self._ck(self.L.gs_get_skin(self.h, C.c_int(i), C.byref(nj), C.byref(joints), C.byref(ibp)))
        m = _view(ibp.value, np.float32, nj.value * 16)
        return _view(joints.value, np.uint32, nj.value), (None if m is None else m.reshape(-1, 16))

    def animation(self, i):
        length, nch = C.c_float(), C.c_int()
        self._ck(self.L.gs_get_animation(self.h, C.c_int(i), C.byref(length), C.byref(nch)))
        return length.value, nch.value

    def channel(self, a, c):
        ci = GsChannelInfo()
        self._ck(self.L.gs_get_channel(self.h, C.c_int(a), C.c_int(c), C.byref(ci)))
        return {"node": ci.node, "path": ci.path, "interpolation": ci.interpolation, "format": ci.format, "width": ci.width,
                "times": _view(ci.times, np.float32, ci.num_times), "transforms": _view(ci.transforms, np.uint8, ci.num_transform_bytes)}

    def sample_channel(self, a, c, time, fix_cubic_spline=False, width=4):
        out = np.zeros(max(width, 64), np.float32)
        self._ck(self.L.gs_sample_channel(self.h, a, c, float(time), int(bool(fix_cubic_spline)), out.ctypes.data_as(C.c_void_p)))
        return out[:width].copy()

    def apply_rest_transforms(self):
        self._ck(self.L.gs_apply_rest_transforms(self.h))

    def animate(self, animation, time):
        self._ck(self.L.gs_animate(self.h, int(animation), float(time)))

    def calculate_global_transforms(self, scene=0):
        self._ck(self.L.gs_calculate_global_transforms(self.h, C.c_int(scene)))

    def player_tick(self, player, dt):
        self._ck(self.L.gs_player_tick(self.h, C.byref(player), float(dt)))

    def gather_lights(self, scene=0):
        n = self._ck(self.L.gs_gather_lights(self.h, C.c_int(scene), None, C.c_int(0)))
        arr = (abi.PtLight * max(n, 1))()
        self._ck(self.L.gs_gather_lights(self.h, C.c_int(scene), arr, C.c_int(n)))
        return list(arr)[:n]

    def gather_bones(self, node):
        n = self._ck(self.L.gs_gather_bones(self.h, C.c_int(node), None, C.c_int(0)))
        arr = (abi.PtBone * max(n, 1))()
        self._ck(self.L.gs_gather_bones(self.h, C.c_int(node), arr, C.c_int(n)))
        return list(arr)[:n]

    def upload(self, renderer):
        """Create the scene's streams / textures / samplers in a Renderer's context (pt_ctx)."""
        self._ck(self.L.gs_upload(self.h, renderer.h))

    def unload(self, renderer):
        """Gltf::Unload: empty the context's instance / material tables and destroy everything upload() created."""
        self._ck(self.L.gs_unload(self.h, renderer.h))

    def frame(self, renderer, scene=0):
        """One frame of host work: skin, gather lights + materials, rebuild the instance table.  Returns light_count."""
        n = C.c_int()
        self._ck(self.L.gs_frame(self.h, renderer.h, C.c_int(scene), C.byref(n)))
        return n.value


def _take(L, ptr, dtype, count):
    buf = (C.c_char * (np.dtype(dtype).itemsize * count)).from_address(ptr.value)
    a = np.frombuffer(buf, dtype=dtype, count=count).copy()
    L.img_free(ptr)
    return a


def decode_rgba8(data):
    """PNG / JPEG bytes -> (H, W, 4) uint8 (the tinygltf image callback, RGBA8)."""
    L = _lib()
    w, h, ptr = C.c_int(), C.c_int(), C.c_void_p()
    b = bytes(data)
    rc = L.img_decode_rgba8(b, len(b), C.byref(w), C.byref(h), C.byref(ptr))
    if rc != 0:
        _err(L, rc)
    return _take(L, ptr, np.uint8, w.value * h.value * 4).reshape(h.value, w.value, 4)


def load_rgba8(path):
    with open(path, "rb") as f:
        return decode_rgba8(f.read())


def decode_rgb32f(data, is_exr):
    """Radiance .hdr / OpenEXR bytes -> ((H, W, 3) float32, half_source) (LoadEnvironmentMapImageHdr / Exr)."""
    L = _lib()
    w, h, half, ptr = C.c_int(), C.c_int(), C.c_int(), C.c_void_p()
    b = bytes(data)
    rc = L.img_decode_rgb32f(b, len(b), int(is_exr), C.byref(w), C.byref(h), C.byref(half), C.byref(ptr))      # 2 = single-channel lookup table
    if rc != 0:
        _err(L, rc)
    return _take(L, ptr, np.float32, w.value * h.value * 3).reshape(h.value, w.value, 3), bool(half.value)


def load_rgb32f(path):
    with open(path, "rb") as f:
        return decode_rgb32f(f.read(), str(path).lower().endswith(".exr"))


def write_png(path, rgba8, channels=3):
    """(H, W, 4) uint8 (pt_tonemap's RGBA8) -> PNG file; channels = 3 drops alpha."""
    L = _lib()
    a = np.ascontiguousarray(rgba8, dtype=np.uint8)
    rc = L.img_write_png(str(path).encode(), a.ctypes.data_as(C.c_void_p), C.c_int(a.shape[1]), C.c_int(a.shape[0]), C.c_int(channels))
    if rc != 0:
        _err(L, rc)


def write_pfm(path, rgb32f):
    L = _lib()
    a = np.ascontiguousarray(rgb32f, dtype=np.float32)
    rc = L.img_write_pfm(str(path).encode(), a.ctypes.data_as(C.c_void_p), C.c_int(a.shape[1]), C.c_int(a.shape[0]))
    if rc != 0:
        _err(L, rc)


def write_exr(path, rgb32f, half=False):
    L = _lib()
    a = np.ascontiguousarray(rgb32f, dtype=np.float32)
    rc = L.img_write_exr(str(path).encode(), a.ctypes.data_as(C.c_void_p), C.c_int(a.shape[1]), C.c_int(a.shape[0]), C.c_int(int(half)))
    if rc != 0:
        _err(L, rc)
