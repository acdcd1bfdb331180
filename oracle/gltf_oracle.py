"""CPU restatement (numpy) of the reference's glTF load path and animation sampler -- TEST INFRASTRUCTURE ONLY
(tests/, never imported by the product).  PARITY UNPINNED: the reference ships no glTF fixtures or tests, and
tinygltf / stb_image / glm are empty submodules, so this restates the reference's own conversion rules:

  accessor conversion   Source/TinyGltfTools.h:137-375 (normalised unpack/pack, plain casts, missing components = 1,
                        sparse substitution, raw copies)
  Gltf::LoadPrimitive   Source/Gltf.cpp:178-319 (stream formats of Source/Mesh.cpp:124-132, u8 -> u16 indices,
                        EncodeTangentSpace / EncodeNormal :65-104, material + 1)
  LoadMaterials         Source/Gltf.cpp:476-633 + GpuMaterial() Source/Renderer.h:125-170
  LoadNodes / globals   Source/Gltf.cpp:654-706, 1016-1041 (Y-up -> Z-up root, T*R*S)
  GatherLights          Source/Renderer.cpp:459-492;  bones Source/Renderer.cpp:408-417
  Animation sampling    Source/Animation.cpp:9-122 (incl. the cubic-spline indexing it marks "I think this is wrong")
Image decode uses PIL here (PNG is lossless: exact; JPEG: compared with a tolerance, see tests/test_gltf_loader.py).
"""
import base64
import io
import json
import math
import os
import struct

import numpy as np

CT = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5124: np.int32, 5125: np.uint32, 5126: np.float32}
NC = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT2": 4, "MAT3": 9, "MAT4": 16}


class Doc:
    """Parsed container: JSON + buffers (GLB chunk, data URIs or files beside the .gltf)."""

    def __init__(self, path):
        self.dir = os.path.dirname(os.path.abspath(path))
        raw = open(path, "rb").read()
        bin_chunk = None
        if path.endswith(".glb"):
            magic, ver, total = struct.unpack_from("<III", raw, 0)
            assert magic == 0x46546C67 and ver == 2
            pos, text = 12, None
            while pos + 8 <= total:
                clen, ctype = struct.unpack_from("<II", raw, pos)
                body = raw[pos + 8:pos + 8 + clen]
                if ctype == 0x4E4F534A and text is None:
                    text = body
                elif ctype == 0x004E4942 and bin_chunk is None:
                    bin_chunk = body
                pos += 8 + clen + ((4 - (clen & 3)) & 3)
            self.j = json.loads(text.decode("utf-8"))
        else:
            self.j = json.loads(raw.decode("utf-8-sig"))
        self.buffers = []
        for i, b in enumerate(self.j.get("buffers", [])):
            if "uri" in b:
                self.buffers.append(self.uri(b["uri"]))
            else:
                self.buffers.append(bin_chunk)

    def uri(self, u):
        if u.startswith("data:"):
            return base64.b64decode(u[u.index(",") + 1:])
        from urllib.parse import unquote
        return open(os.path.join(self.dir, unquote(u)), "rb").read()

    def view_bytes(self, v):
        bv = self.j["bufferViews"][v]
        off = bv.get("byteOffset", 0)
        return self.buffers[bv["buffer"]][off:off + bv["byteLength"]], bv.get("byteStride", 0)

    def elements(self, acc_index):
        """(count, ncomp) array of the accessor's raw components in their own dtype, sparse applied; None rows where the
        reference reads a null base are zeros (TinyGltfTools.h:195-199)."""
        a = self.j["accessors"][acc_index]
        dt, nc, count = np.dtype(CT[a["componentType"]]), NC[a["type"]], a["count"]
        out = np.zeros((count, nc), dt)
        if "bufferView" in a:
            data, stride = self.view_bytes(a["bufferView"])
            stride = stride or dt.itemsize * nc
            base = a.get("byteOffset", 0)
            for i in range(count):
                out[i] = np.frombuffer(data, dt, nc, base + i * stride)
        sp = a.get("sparse")
        if sp:
            idata, istride = self.view_bytes(sp["indices"]["bufferView"])
            idt = np.dtype(CT[sp["indices"]["componentType"]])
            vdata, vstride = self.view_bytes(sp["values"]["bufferView"])
            for k in range(sp["count"]):
                idx = int(np.frombuffer(idata, idt, 1, sp["indices"].get("byteOffset", 0) + k * (istride or idt.itemsize))[0])
                out[idx] = np.frombuffer(vdata, dt, nc, sp["values"].get("byteOffset", 0) + k * (vstride or dt.itemsize * nc))
        return out, a


def unpack_normalized(x):
    """UnpackNormalizedValue (TinyGltfTools.h:137-158) on an array in its stored dtype -> float32."""
    dt = x.dtype
    if dt == np.float32:
        return x
    if dt == np.uint8:
        return x.astype(np.float32) / np.float32(255)
    if dt == np.int8:
        return np.clip(x.astype(np.float32) / np.float32(127), -1, 1).astype(np.float32)
    if dt == np.uint16:
        return x.astype(np.float32) / np.float32(65535)
    if dt == np.int16:
        return np.clip(x.astype(np.float32) / np.float32(32767), -1, 1).astype(np.float32)
    if dt == np.uint32:
        return x.astype(np.float32) / np.float32(4294967295.0)
    return np.clip(x.astype(np.float32) / np.float32(2147483647.0), -1, 1).astype(np.float32)


def c_round(x):
    return np.where(x >= 0, np.floor(x + np.float32(0.5)), np.ceil(x - np.float32(0.5))).astype(np.float32)


def pack_normalized(f, dtype):
    """PackNormalizedValue :160-171 (glm::packUnorm / packSnorm = round(clamp(x) * max))."""
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return f.astype(np.float32)
    info = np.iinfo(dtype)
    lo = np.float32(0 if info.min == 0 else -1)
    return c_round(np.clip(f.astype(np.float32), lo, np.float32(1)) * np.float32(info.max)).astype(dtype)


def convert(doc, acc_index, L, dtype, normalize=False):
    """Copy<L, T, NORMALIZE> (TinyGltfTools.h:340-356 via Convert :195-222)."""
    raw, a = doc.elements(acc_index)
    dtype = np.dtype(dtype)
    n = min(L, raw.shape[1])
    out = np.ones((raw.shape[0], L), dtype)                    # missing components = (T)1
    src = raw[:, :n]
    if raw.dtype == dtype:
        out[:, :n] = src
    elif a.get("normalized", False) or normalize:
        out[:, :n] = pack_normalized(unpack_normalized(src), dtype)
    else:
        with np.errstate(all="ignore"):
            out[:, :n] = src.astype(dtype)                     # C cast
    return out


def load_primitive(doc, gp, encode_tangent_space, encode_normal):
    mode = gp.get("mode", 4)
    attrs = gp["attributes"]
    p = {"topology": mode}
    if mode in (2, 6):
        p["valid"] = False
        return p
    flags = 0
    if gp.get("indices", -1) != -1:
        flags |= 1
    for bit, name in ((2, "NORMAL"), (4, "TEXCOORD_0"), (8, "TEXCOORD_1"), (16, "COLOR_0")):
        if name in attrs:
            flags |= bit
    if "JOINTS_0" in attrs and "WEIGHTS_0" in attrs:
        flags |= 32
    p["flags"] = flags
    p["num_vertices"] = doc.j["accessors"][attrs["POSITION"]]["count"]
    p["index"] = None
    if flags & 1:
        raw, a = doc.elements(gp["indices"])
        p["index"] = raw[:, 0].astype(np.uint16) if a["componentType"] in (5121, 5123) else raw[:, 0].astype(np.uint32)
    p["position"] = convert(doc, attrs["POSITION"], 3, np.float32)
    p["tangent_space"] = None
    if flags & 2:
        n = convert(doc, attrs["NORMAL"], 3, np.float32)
        if "TANGENT" in attrs:
            t = convert(doc, attrs["TANGENT"], 4, np.float32)
            m = min(len(n), len(t))
            p["tangent_space"] = encode_tangent_space(n[:m], t[:m])
        else:
            p["tangent_space"] = encode_normal(n)
    p["texcoord0"] = convert(doc, attrs["TEXCOORD_0"], 2, np.float32) if flags & 4 else None
    p["texcoord1"] = convert(doc, attrs["TEXCOORD_1"], 2, np.float32) if flags & 8 else None
    p["color"] = convert(doc, attrs["COLOR_0"], 4, np.uint16, normalize=True) if flags & 16 else None
    p["joint_weight"] = None
    if flags & 32:
        j = convert(doc, attrs["JOINTS_0"], 4, np.uint16)
        w = convert(doc, attrs["WEIGHTS_0"], 4, np.uint16, normalize=True)
        p["joint_weight"] = np.concatenate([j, w], axis=1)
    p["material_id"] = gp.get("material", -1) + 1
    p["targets"] = []
    for tg in gp.get("targets", []):
        t = {"position": convert(doc, tg["POSITION"], 3, np.float32) if "POSITION" in tg else None, "tangent_space": None}
        if "NORMAL" in tg:
            n = convert(doc, tg["NORMAL"], 3, np.float32)
            t["tangent_space"] = encode_tangent_space(n, convert(doc, tg["TANGENT"], 4, np.float32)) if "TANGENT" in tg else encode_normal(n)
        p["targets"].append(t)
    return p


# ---- materials --------------------------------------------------------------------------------------------------
def _tex_default():
    return {"texture": -1, "sampler": 0, "tex_coord": 0, "offset": (0.0, 0.0), "scale": (1.0, 1.0), "rotation": 0.0}


class MaterialLoader:
    def __init__(self, doc):
        self.doc = doc
        self.first_use_srgb = {}           # image index -> sRGB flag of the first reference (Gltf.cpp:409-412)

    def get_texture(self, index, tex_coord, transform, srgb):
        t = _tex_default()
        if index == -1:
            return t
        tex = self.doc.j["textures"][index]
        source = tex.get("source", -1)
        if source == -1:
            return t
        self.first_use_srgb.setdefault(source, srgb)
        t["texture"] = source
        t["sampler"] = 0 if tex.get("sampler", -1) == -1 else tex["sampler"] + 1
        t["tex_coord"] = tex_coord if 0 <= tex_coord < 2 else 0
        if isinstance(transform, dict):
            if len(transform.get("offset", [])) == 2:
                t["offset"] = tuple(np.float32(v) for v in transform["offset"])
            if isinstance(transform.get("rotation"), (int, float)):
                t["rotation"] = np.float32(transform["rotation"])
            if len(transform.get("scale", [])) == 2:
                t["scale"] = tuple(np.float32(v) for v in transform["scale"])
            tc = transform.get("texCoord")
            if isinstance(tc, int) and 0 <= tc < 2:
                t["tex_coord"] = tc
        return t

    def core(self, info, srgb):
        if not isinstance(info, dict):
            return _tex_default()
        return self.get_texture(info.get("index", -1), info.get("texCoord", 0), info.get("extensions", {}).get("KHR_texture_transform"), srgb)

    def ext(self, info, srgb):
        if not isinstance(info, dict):
            return _tex_default()
        return self.get_texture(int(info.get("index", 0)), int(info.get("texCoord", 0)), info.get("extensions", {}).get("KHR_texture_transform"), srgb)

    def load(self):
        """list of dicts with the GpuMaterial field names (Renderer.h:88-171); index 0 = default material."""
        mats = [self.gpu(self.defaults())]
        for gm in self.doc.j.get("materials", []):
            m = self.defaults()
            pbr = gm.get("pbrMetallicRoughness", {})
            nt = gm.get("normalTexture")
            m["normal"] = self.core(nt, False)
            if isinstance(nt, dict) and "scale" in nt:
                m["normal_scale"] = nt["scale"]
            m["albedo"] = self.core(pbr.get("baseColorTexture"), True)
            m["base_color_factor"] = pbr.get("baseColorFactor", [1, 1, 1, 1])
            m["metallic_roughness"] = self.core(pbr.get("metallicRoughnessTexture"), False)
            m["metalness_factor"] = pbr.get("metallicFactor", 1.0)
            m["roughness_factor"] = pbr.get("roughnessFactor", 1.0)
            m["occlusion"] = self.core(gm.get("occlusionTexture"), False)
            m["emissive"] = self.core(gm.get("emissiveTexture"), True)
            m["emissive_factor"] = gm.get("emissiveFactor", [0, 0, 0])
            m["alpha_mode"] = {"OPAQUE": 0, "MASK": 1, "BLEND": 2}.get(gm.get("alphaMode", "OPAQUE"), 0)
            m["alpha_cutoff"] = gm.get("alphaCutoff", 0.5)
            if gm.get("doubleSided"):
                m["flags"] |= 1
            ex = gm.get("extensions", {})
            e = ex.get("KHR_materials_anisotropy")
            if e is not None:
                m["anisotropy_strength"] = e.get("anisotropyStrength", m["anisotropy_strength"])
                m["anisotropy_rotation"] = e.get("anisotropyRotation", m["anisotropy_rotation"])
                m["anisotropy"] = self.ext(e.get("anisotropyTexture"), False)
            e = ex.get("KHR_materials_clearcoat")
            if e is not None:
                m["clearcoat_factor"] = e.get("clearcoatFactor", 0.0)
                m["clearcoat_roughness_factor"] = e.get("clearcoatRoughnessFactor", 0.0)
                m["clearcoat"] = self.ext(e.get("clearcoatTexture"), False)
                m["clearcoat_roughness"] = self.ext(e.get("clearcoatRoughnessTexture"), False)
                cn = e.get("clearcoatNormalTexture")
                m["clearcoat_normal"] = self.ext(cn, False)
                if isinstance(cn, dict) and "scale" in cn:
                    m["clearcoat_normal_scale"] = cn["scale"]
            e = ex.get("KHR_materials_emissive_strength")
            if e is not None:
                m["emissive_strength"] = e.get("emissiveStrength", 1.0)
            e = ex.get("KHR_materials_ior")
            if e is not None:
                m["ior"] = e.get("ior", 1.5)
            e = ex.get("KHR_materials_sheen")
            if e is not None:
                if len(e.get("sheenColorFactor", [])) == 3:
                    m["sheen_color_factor"] = e["sheenColorFactor"]
                m["sheen_roughness_factor"] = e.get("sheenRoughnessFactor", 0.0)
                m["sheen_color"] = self.ext(e.get("sheenColorTexture"), True)
                m["sheen_roughness"] = self.ext(e.get("sheenRoughnessTexture"), False)
            e = ex.get("KHR_materials_specular")
            if e is not None:
                m["specular_factor"] = e.get("specularFactor", 1.0)
                if len(e.get("specularColorFactor", [])) == 3:
                    m["specular_color_factor"] = e["specularColorFactor"]
                m["specular"] = self.ext(e.get("specularTexture"), False)
                m["specular_color"] = self.ext(e.get("specularColorTexture"), True)
            e = ex.get("KHR_materials_transmission")
            if e is not None:
                m["transmission_factor"] = e.get("transmissionFactor", 0.0)
                m["transmission"] = self.ext(e.get("transmissionTexture"), False)
            e = ex.get("KHR_materials_volume")
            if e is not None:
                m["thickness_factor"] = e.get("thicknessFactor", 0.0)
                m["thickness"] = self.ext(e.get("thicknessTexture"), False)
                m["attenuation_distance"] = e.get("attenuationDistance", 0.0)
                if len(e.get("attenuationColor", [])) == 3:
                    m["attenuation_color"] = e["attenuationColor"]
            if "KHR_materials_unlit" in ex:
                m["flags"] |= 2
            mats.append(self.gpu(m))
        return mats

    @staticmethod
    def defaults():
        d = {"flags": 0, "alpha_mode": 0, "metalness_factor": 1.0, "roughness_factor": 1.0, "base_color_factor": [1, 1, 1, 1], "occlusion_factor": 1.0,
             "emissive_factor": [0, 0, 0], "emissive_strength": 1.0, "alpha_cutoff": 0.5, "ior": 1.5, "normal_scale": 1.0, "specular_factor": 1.0,
             "specular_color_factor": [1, 1, 1], "clearcoat_factor": 0.0, "clearcoat_roughness_factor": 0.0, "clearcoat_normal_scale": 1.0,
             "anisotropy_strength": 0.0, "anisotropy_rotation": 0.0, "sheen_color_factor": [0, 0, 0], "sheen_roughness_factor": 0.0, "transmission_factor": 0.0,
             "thickness_factor": 0.0, "attenuation_distance": 0.0, "attenuation_color": [1, 1, 1]}
        for k in ("normal", "albedo", "metallic_roughness", "occlusion", "emissive", "specular", "specular_color", "clearcoat", "clearcoat_roughness",
                  "clearcoat_normal", "anisotropy", "sheen_color", "sheen_roughness", "transmission", "thickness"):
            d[k] = _tex_default()
        return d

    @staticmethod
    def gpu(m):
        g = dict(m)
        s = np.float32(m["emissive_strength"])
        g["emissive_factor"] = [np.float32(s * np.float32(v)) for v in m["emissive_factor"]]
        g["alpha_cutoff"] = m["alpha_cutoff"] if m["alpha_mode"] == 1 else 0.0
        return g


# ---- nodes, transforms, lights, bones --------------------------------------------------------------------------
def quat_to_mat(q):
    x, y, z, w = [np.float32(v) for v in q]
    two = np.float32(2)
    one = np.float32(1)
    m = np.eye(4, dtype=np.float32)
    m[0, 0] = one - two * (y * y + z * z); m[1, 0] = two * (x * y + w * z); m[2, 0] = two * (x * z - w * y)
    m[0, 1] = two * (x * y - w * z); m[1, 1] = one - two * (x * x + z * z); m[2, 1] = two * (y * z + w * x)
    m[0, 2] = two * (x * z + w * y); m[1, 2] = two * (y * z - w * x); m[2, 2] = one - two * (x * x + y * y)
    return m


def matmul32(a, b):
    """fp32 product accumulated k = 0..3 in order (glm's operator*)."""
    r = np.zeros((4, 4), np.float32)
    for c in range(4):
        for row in range(4):
            s = np.float32(0)
            for k in range(4):
                s = np.float32(s + np.float32(a[row, k] * b[k, c]))
            r[row, c] = s
    return r


def local_matrix(t, r, s):
    T = np.eye(4, dtype=np.float32); T[:3, 3] = np.asarray(t, np.float32)
    S = np.diag(np.asarray(list(s) + [1], np.float32)).astype(np.float32)
    return T, quat_to_mat(r), S


def global_transforms(doc, scene=0, locals_override=None):
    """{node: 4x4 float32}: CalculateGlobalTransforms (Gltf.cpp:1016-1041) with the Y-up -> Z-up root."""
    nodes = doc.j.get("nodes", [])
    cs = np.array([[1, 0, 0, 0], [0, 0, -1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], np.float32)
    out = {}

    def trs_of(i):
        if locals_override and i in locals_override:
            return locals_override[i]
        n = nodes[i]
        return (n.get("translation", [0, 0, 0]), n.get("rotation", [0, 0, 0, 1]), n.get("scale", [1, 1, 1]))

    def walk(i, parent):
        t, r, s = trs_of(i)
        T, R, S = local_matrix(t, r, s)
        g = matmul32(matmul32(matmul32(parent, T), R), S)
        out[i] = g
        for c in nodes[i].get("children", []):
            walk(c, g)

    for root in doc.j["scenes"][scene]["nodes"]:
        walk(root, cs)
    return out


def traversal_order(doc, scene=0):
    order = []

    def walk(i):
        order.append(i)
        for c in doc.j["nodes"][i].get("children", []):
            walk(c)
    for r in doc.j["scenes"][scene]["nodes"]:
        walk(r)
    return order


def gather_lights(doc, globals_, scene=0):
    lights = doc.j.get("extensions", {}).get("KHR_lights_punctual", {}).get("lights", [])
    out = []
    for i in traversal_order(doc, scene):
        li = doc.j["nodes"][i].get("extensions", {}).get("KHR_lights_punctual", {}).get("light", -1)
        if li == -1:
            continue
        L, g = lights[li], globals_[i]
        it = np.linalg.inv(g.astype(np.float64)).T
        d = it @ np.array([0, 0, -1, 0], np.float64)
        d = d / np.linalg.norm(d)
        out.append({"type": {"point": 0, "spot": 1, "directional": 2}[L["type"]], "position": g[:3, 3].copy(), "cutoff": L.get("range", 0.0),
                    "direction": d[:3].astype(np.float32), "intensity": L.get("intensity", 1.0), "color": L.get("color", [1, 1, 1]),
                    "inner_angle": L.get("spot", {}).get("innerConeAngle", 0.0), "outer_angle": L.get("spot", {}).get("outerConeAngle", math.pi / 4)})
    return out


def gather_bones(doc, globals_, node):
    skin = doc.j["skins"][doc.j["nodes"][node]["skin"]]
    if "inverseBindMatrices" in skin:
        raw, _ = doc.elements(skin["inverseBindMatrices"])
        ibm = [raw[k].reshape(4, 4).T.astype(np.float32) for k in range(len(raw))]       # column-major storage
    else:
        ibm = [np.eye(4, dtype=np.float32)] * len(skin["joints"])
    ninv = np.linalg.inv(globals_[node].astype(np.float64))
    out = []
    for k, j in enumerate(skin["joints"]):
        t = ninv @ globals_[j].astype(np.float64) @ ibm[k].astype(np.float64)
        it = np.eye(4)
        it[:3, :3] = np.linalg.inv(t[:3, :3]).T
        out.append((t.astype(np.float32), it.astype(np.float32)))
    return out


# ---- animation (Animation.cpp) -----------------------------------------------------------------------------------
def _unpack_component(fmt, raw):
    if fmt == 0:
        return np.float32(raw)
    if fmt == 2:
        return np.float32(raw) / np.float32(65535)
    if fmt == 1:
        return np.float32(raw) / np.float32(255)
    return max(np.float32(-1), min(np.float32(1), np.float32(raw) / np.float32(127)))    # SNORM_8 (also what SHORT maps to, Gltf.cpp:767)


def load_channel(doc, anim, ch):
    smp = anim["samplers"][ch["sampler"]]
    path = {"translation": 0, "rotation": 1, "scale": 2, "weights": 3}[ch["target"]["path"]]
    times = convert(doc, smp["input"], 1, np.float32)[:, 0]
    raw, a = doc.elements(smp["output"])
    fmt = {5126: 0, 5123: 2, 5122: 3, 5121: 1, 5120: 3}[a["componentType"]]
    node = ch["target"]["node"]
    if path == 3:
        mesh = doc.j["nodes"][node]["mesh"]
        width = len(doc.j["meshes"][mesh]["primitives"][0].get("targets", []))
    else:
        width = 4 if path == 1 else 3
    return {"node": node, "path": path, "interp": {"STEP": 0, "LINEAR": 1, "CUBICSPLINE": 2}[smp.get("interpolation", "LINEAR")], "format": fmt,
            "width": width, "times": times, "bytes": raw.tobytes(), "dtype": raw.dtype,
            "end": doc.j["accessors"][smp["input"]].get("max", [float(times[-1])])[0]}


def _data(c, keyframe, comp):
    """UnpackData (Animation.cpp:52-71): byte offset keyframe*width*size + comp*size into the raw output stream.  SHORT data is
    read through the SNORM_8 path the loader assigns it (first byte, as int8)."""
    fs = {0: 4, 2: 2, 4: 2, 1: 1, 3: 1}[c["format"]]
    off = keyframe * c["width"] * fs + comp * fs
    b = c["bytes"]
    if off + fs > len(b):
        return np.float32(0)
    if c["format"] == 0:
        return np.frombuffer(b, np.float32, 1, off)[0]
    if c["format"] == 2:
        return _unpack_component(2, np.frombuffer(b, np.uint16, 1, off)[0])
    if c["format"] == 1:
        return _unpack_component(1, b[off])
    return _unpack_component(3, np.frombuffer(b, np.int8, 1, off)[0])


def slerp(a, b, t):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    cos = np.float32(np.dot(a, b))
    if cos < 0:
        b, cos = -b, -cos
    t = np.float32(t)
    if cos > np.float32(1) - np.float32(1.1920929e-07):
        return (a + t * (b - a)).astype(np.float32)
    ang = np.float32(math.acos(cos))
    return ((np.float32(math.sin((1 - t) * ang)) * a + np.float32(math.sin(t * ang)) * b) / np.float32(math.sin(ang))).astype(np.float32)


def sample_channel(c, time, fix_cubic=False):
    times = c["times"]
    time = np.float32(min(max(np.float32(time), times[0]), times[-1]))
    ks = 0
    for i in range(1, len(times)):
        if times[i] <= time:
            ks = i
        else:
            break
    ke = ks
    if ke + 1 < len(times) and times[ke] < time:
        ke += 1
    diff = np.float32(times[ke] - times[ks])
    f = np.float32(0) if diff == 0 else np.float32((time - times[ks]) / diff)
    w = c["width"]
    if c["interp"] == 0:
        return np.array([_data(c, ks, i) for i in range(w)], np.float32)
    if c["interp"] == 1:
        if c["path"] == 1:
            return slerp([_data(c, ks, i) for i in range(4)], [_data(c, ke, i) for i in range(4)], f)
        out = []
        for i in range(w):
            a, b = _data(c, ks, i), _data(c, ke, i)
            out.append(np.float32(f * b + (np.float32(1) - f) * a) if (a <= 0 <= b or a >= 0 >= b) else (b if f == 1 else np.float32(a + f * (b - a))))
        return np.array(out, np.float32)
    out = []
    for i in range(w):
        if not fix_cubic:
            sv = st = _data(c, ks * 3, i)
            ev = et = _data(c, ke * 3, i)
        else:
            sv, st, ev, et = _data(c, ks * 3 + 1, i), _data(c, ks * 3 + 2, i), _data(c, ke * 3 + 1, i), _data(c, ke * 3, i)
        t = f; t2 = np.float32(t * t); t3 = np.float32(t2 * t)
        out.append(np.float32((2 * t3 - 3 * t2 + 1) * sv + diff * (t3 - 2 * t2 + t) * st + (-2 * t3 + 3 * t2) * ev + diff * (t3 - t2) * et))
    out = np.array(out, np.float32)
    if c["path"] == 1:
        out = out / np.float32(np.linalg.norm(out))
    return out


def decode_image(doc, image_index):
    """RGBA8 pixels of images[i] via PIL (oracle side only)."""
    from PIL import Image
    img = doc.j["images"][image_index]
    if "uri" in img:
        data = doc.uri(img["uri"])
    else:
        data, _ = doc.view_bytes(img["bufferView"])
    im = Image.open(io.BytesIO(data))
    if im.mode.startswith("I"):                               # 16-bit grey: keep the high byte (stb_image's 16 -> 8 reduction)
        g = (np.asarray(im).astype(np.uint32) >> 8).astype(np.uint8)
        return np.stack([g, g, g, np.full(g.shape, 255, np.uint8)], -1)
    return np.asarray(im.convert("RGBA"))
