"""Minimal glTF 2.0 / GLB writer for the loader tests: builds the JSON + binary blob from numpy arrays with control over the
things the reference's conversion rules care about (component type, normalised flag, byteStride / interleaving, sparse
accessors, embedded vs external resources)."""
import base64
import json
import os
import struct

import numpy as np

COMPONENT = {np.dtype(np.int8): 5120, np.dtype(np.uint8): 5121, np.dtype(np.int16): 5122, np.dtype(np.uint16): 5123, np.dtype(np.int32): 5124,
             np.dtype(np.uint32): 5125, np.dtype(np.float32): 5126}
TYPE = {1: "SCALAR", 2: "VEC2", 3: "VEC3", 4: "VEC4", 16: "MAT4"}


class Builder:
    def __init__(self):
        self.j = {"asset": {"version": "2.0", "generator": "tests/gltf_writer.py"}, "buffers": [], "bufferViews": [], "accessors": [], "meshes": [], "nodes": [],
                  "scenes": [{"nodes": []}], "scene": 0}
        self.blob = bytearray()
        self.image_files = {}            # name -> bytes (external mode)

    # ---- binary
    def view(self, data, stride=0, align=4):
        while len(self.blob) % align:
            self.blob.append(0)
        off = len(self.blob)
        self.blob += bytes(data)
        v = {"buffer": 0, "byteOffset": off, "byteLength": len(bytes(data))}
        if stride:
            v["byteStride"] = stride
        self.j["bufferViews"].append(v)
        return len(self.j["bufferViews"]) - 1

    def accessor(self, arr, normalized=False, stride=0, minmax=False, type_override=None):
        """arr: (count, ncomp) or (count,) numpy array; stride > element size pads every element."""
        a = np.ascontiguousarray(arr)
        if a.ndim == 1:
            a = a[:, None]
        count, nc = a.shape
        es = a.dtype.itemsize * nc
        if stride and stride > es:
            raw = bytearray(count * stride)
            for i in range(count):
                raw[i * stride:i * stride + es] = a[i].tobytes()
            v = self.view(raw, stride)
        else:
            v = self.view(a.tobytes())
        acc = {"bufferView": v, "componentType": COMPONENT[a.dtype], "count": int(count), "type": type_override or TYPE[nc]}
        if normalized:
            acc["normalized"] = True
        if minmax:
            acc["min"] = [float(x) for x in a.min(axis=0)]
            acc["max"] = [float(x) for x in a.max(axis=0)]
        self.j["accessors"].append(acc)
        return len(self.j["accessors"]) - 1

    def interleaved(self, arrays, normalized=None):
        """Several (count, nc) arrays sharing one strided bufferView; returns their accessor indices."""
        arrays = [np.ascontiguousarray(a if a.ndim == 2 else a[:, None]) for a in arrays]
        count = arrays[0].shape[0]
        sizes = [a.dtype.itemsize * a.shape[1] for a in arrays]
        stride = (sum(sizes) + 3) & ~3
        raw = bytearray(count * stride)
        offs, o = [], 0
        for a, sz in zip(arrays, sizes):
            offs.append(o)
            for i in range(count):
                raw[i * stride + o:i * stride + o + sz] = a[i].tobytes()
            o += sz
        v = self.view(raw, stride)
        out = []
        for k, (a, o) in enumerate(zip(arrays, offs)):
            acc = {"bufferView": v, "byteOffset": o, "componentType": COMPONENT[a.dtype], "count": int(count), "type": TYPE[a.shape[1]]}
            if normalized and normalized[k]:
                acc["normalized"] = True
            self.j["accessors"].append(acc)
            out.append(len(self.j["accessors"]) - 1)
        return out

    def sparse_accessor(self, base, indices, values, with_base=True):
        """accessor = base with `values` substituted at `indices` (index dtype decides the sparse index component type)."""
        base = np.ascontiguousarray(base)
        acc_i = self.accessor(base) if with_base else None
        acc = dict(self.j["accessors"][acc_i]) if with_base else {"componentType": COMPONENT[base.dtype], "count": int(base.shape[0]), "type": TYPE[base.shape[1]]}
        iv = self.view(np.ascontiguousarray(indices).tobytes())
        vv = self.view(np.ascontiguousarray(values).tobytes())
        acc["sparse"] = {"count": int(len(indices)), "indices": {"bufferView": iv, "componentType": COMPONENT[np.asarray(indices).dtype]}, "values": {"bufferView": vv}}
        self.j["accessors"].append(acc)
        return len(self.j["accessors"]) - 1

    # ---- images / textures
    def image(self, data, mime, name, embed="view"):
        self.j.setdefault("images", [])
        if embed == "view":
            self.j["images"].append({"bufferView": self.view(data), "mimeType": mime, "name": name})
        elif embed == "uri":
            self.j["images"].append({"uri": "data:%s;base64,%s" % (mime, base64.b64encode(data).decode()), "name": name})
        else:
            fn = name + (".png" if mime == "image/png" else ".jpg")
            self.image_files[fn] = bytes(data)
            self.j["images"].append({"uri": fn, "name": name})
        return len(self.j["images"]) - 1

    def texture(self, source, sampler=None):
        self.j.setdefault("textures", [])
        t = {"source": source}
        if sampler is not None:
            t["sampler"] = sampler
        self.j["textures"].append(t)
        return len(self.j["textures"]) - 1

    def sampler(self, **kw):
        self.j.setdefault("samplers", []).append(kw)
        return len(self.j["samplers"]) - 1

    def material(self, m):
        self.j.setdefault("materials", []).append(m)
        return len(self.j["materials"]) - 1

    def mesh(self, primitives, weights=None, name=""):
        m = {"primitives": primitives, "name": name}
        if weights is not None:
            m["weights"] = weights
        self.j["meshes"].append(m)
        return len(self.j["meshes"]) - 1

    def node(self, root=False, **kw):
        self.j["nodes"].append(kw)
        i = len(self.j["nodes"]) - 1
        if root:
            self.j["scenes"][0]["nodes"].append(i)
        return i

    # ---- output
    def _finish(self):
        j = json.loads(json.dumps(self.j))
        j["buffers"] = [{"byteLength": len(self.blob)}]
        return j

    def write_glb(self, path):
        j = self._finish()
        text = json.dumps(j, separators=(",", ":")).encode()
        text += b" " * ((4 - len(text) % 4) % 4)
        blob = bytes(self.blob) + b"\0" * ((4 - len(self.blob) % 4) % 4)
        with open(path, "wb") as f:
            f.write(struct.pack("<III", 0x46546C67, 2, 12 + 8 + len(text) + 8 + len(blob)))
            f.write(struct.pack("<II", len(text), 0x4E4F534A) + text)
            f.write(struct.pack("<II", len(blob), 0x004E4942) + blob)
        return path

    def write_gltf(self, path, external_bin=False):
        j = self._finish()
        if external_bin:
            bn = os.path.splitext(os.path.basename(path))[0] + ".bin"
            open(os.path.join(os.path.dirname(path), bn), "wb").write(bytes(self.blob))
            j["buffers"][0]["uri"] = bn
        else:
            j["buffers"][0]["uri"] = "data:application/octet-stream;base64," + base64.b64encode(bytes(self.blob)).decode()
        for fn, data in self.image_files.items():
            open(os.path.join(os.path.dirname(path), fn), "wb").write(data)
        with open(path, "w") as f:
            json.dump(j, f, indent=1)
        return path
