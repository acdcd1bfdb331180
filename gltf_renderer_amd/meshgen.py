"""Procedural geometry + the reference's CPU-side vertex packing (host logic).

Packing follows Source/Gltf.cpp:23-104 (EncodeOctahedralMap, CreateBasis, EncodeNormal,
EncodeTangentSpace) and the stream formats of Source/Mesh.cpp:124-132 / Source/Gltf.cpp:242-311:
index u16/u32, position f32x3, tangent space R10G10B10A2, texcoord f32x2, colour unorm16x4,
joints u16x4 + weights unorm16x4.  All arithmetic in float32 like the C++.
"""
import numpy as np

f32 = np.float32


def _norm(v):
    v = np.asarray(v, dtype=np.float64)
    n = np.linalg.norm(v, axis=-1, keepdims=True)
    return v / np.maximum(n, 1e-30)


# ---- Gltf.cpp:23-63 ------------------------------------------------------------------------------
def encode_octahedral(n):
    n = n.astype(f32)
    o = n / (np.abs(n[:, 0:1]) + np.abs(n[:, 1:2]) + np.abs(n[:, 2:3]))
    sx = np.where(o[:, 0] >= 0, f32(1), f32(-1))
    sy = np.where(o[:, 1] >= 0, f32(1), f32(-1))
    fx = sx * (f32(1) - np.abs(o[:, 1]))
    fy = sy * (f32(1) - np.abs(o[:, 0]))
    up = o[:, 2] >= 0
    return np.stack([np.where(up, o[:, 0], fx), np.where(up, o[:, 1], fy)], axis=1).astype(f32)


def decode_octahedral(e):
    e = e.astype(f32)
    z = f32(1) - np.abs(e[:, 0]) - np.abs(e[:, 1])
    sx = np.where(e[:, 0] >= 0, f32(1), f32(-1))
    sy = np.where(e[:, 1] >= 0, f32(1), f32(-1))
    x = np.where(z >= 0, e[:, 0], sx * (f32(1) - np.abs(e[:, 1])))
    y = np.where(z >= 0, e[:, 1], sy * (f32(1) - np.abs(e[:, 0])))
    r = np.stack([x, y, z], axis=1).astype(f32)
    return (r / np.sqrt((r * r).sum(axis=1, keepdims=True, dtype=f32))).astype(f32)


def create_basis(n):
    n = n.astype(f32)
    sign = np.where(n[:, 2] >= 0, f32(1), f32(-1))
    a = f32(-1) / (sign + n[:, 2])
    b = n[:, 0] * n[:, 1] * a
    t = np.stack([f32(1) + sign * n[:, 0] * n[:, 0] * a, sign * b, -sign * n[:, 0]], axis=1)
    bt = np.stack([b, sign + n[:, 1] * n[:, 1] * a, -n[:, 1]], axis=1)
    return t.astype(f32), bt.astype(f32)


def _quantize_normal(normal):
    en = f32(0.5) * encode_octahedral(normal) + f32(0.5)
    q = (np.clip(en, 0, 1).astype(f32) * f32(1023.0) + f32(0.5)).astype(np.uint32)
    return q


def encode_normal(normal):
    """Gltf.cpp:65-77: tangent angle bits 0, winding 3."""
    q = _quantize_normal(np.asarray(normal, dtype=f32))
    return (q[:, 0] | (q[:, 1] << 10) | (np.uint32(3) << 30)).astype(np.uint32)


def encode_tangent_space(normal, tangent4):
    """Gltf.cpp:79-104."""
    normal = np.asarray(normal, dtype=f32)
    tangent4 = np.asarray(tangent4, dtype=f32)
    q = _quantize_normal(normal)
    un = q.astype(f32) / f32(1023.0)
    nq = decode_octahedral(f32(2.0) * un - f32(1.0))
    ct, cb = create_basis(nq)
    t3 = tangent4[:, :3]
    angle = np.arctan2((t3 * cb).sum(axis=1, dtype=f32), (t3 * ct).sum(axis=1, dtype=f32)).astype(f32)
    enc = (angle / f32(6.283185307179586) + f32(0.5)).astype(f32)
    qt = (np.clip(enc, 0, 1).astype(f32) * f32(1023.0) + f32(0.5)).astype(np.uint32)
    qw = np.where(tangent4[:, 3] == 1.0, np.uint32(3), np.uint32(0))
    return (q[:, 0] | (q[:, 1] << 10) | (qt << 20) | (qw << 30)).astype(np.uint32)


def pack_unorm16(x):
    """glm::packUnorm<uint16_t>: round(clamp(x,0,1)*65535)."""
    return np.round(np.clip(np.asarray(x, dtype=f32), 0, 1) * f32(65535.0)).astype(np.uint16)


def pack_joint_weight(joints, weights):
    """16 B per vertex {u16 x4 joints, unorm16 x4 weights} (Gltf.cpp:301-311)."""
    j = np.asarray(joints, dtype=np.uint16)
    w = pack_unorm16(weights)
    return np.concatenate([j, w], axis=1).astype(np.uint16)


# ---- meshes ----------------------------------------------------------------------------------------
class Mesh:
    def __init__(self, positions, indices, normals=None, tangents=None, uv0=None, uv1=None, colors=None,
                 joints=None, weights=None):
        self.positions = np.ascontiguousarray(positions, dtype=f32)
        self.indices = None if indices is None else np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
        self.normals = None if normals is None else np.ascontiguousarray(normals, dtype=f32)
        self.tangents = None if tangents is None else np.ascontiguousarray(tangents, dtype=f32)
        self.uv0 = None if uv0 is None else np.ascontiguousarray(uv0, dtype=f32)
        self.uv1 = None if uv1 is None else np.ascontiguousarray(uv1, dtype=f32)
        self.colors = None if colors is None else np.ascontiguousarray(colors, dtype=f32)
        self.joints = joints
        self.weights = weights

    @property
    def num_vertices(self):
        return len(self.positions)

    @property
    def num_indices(self):
        return len(self.indices) if self.indices is not None else len(self.positions)

    def tangent_space_stream(self):
        if self.normals is None:
            return None
        if self.tangents is None:
            return encode_normal(self.normals)
        return encode_tangent_space(self.normals, self.tangents)

    def index_stream(self):
        if self.indices is None:
            return None, None
        from . import abi
        if self.num_vertices <= 65535:
            return self.indices.astype(np.uint16), abi.FORMAT_R16_UINT
        return self.indices.astype(np.uint32), abi.FORMAT_R32_UINT


def grid(nx, ny, origin, du, dv, uv_scale=(1.0, 1.0), displace=None):
    """Tessellated parallelogram: origin + s*du + t*dv, s,t in [0,1]; normal = normalize(du x dv)."""
    origin, du, dv = (np.asarray(a, dtype=np.float64) for a in (origin, du, dv))
    s, t = np.meshgrid(np.linspace(0, 1, nx + 1), np.linspace(0, 1, ny + 1), indexing="xy")
    s, t = s.reshape(-1), t.reshape(-1)
    n = _norm(np.cross(du, dv))
    p = origin[None] + s[:, None] * du[None] + t[:, None] * dv[None]
    if displace is not None:
        p = p + displace(s, t)[:, None] * n[None]
    normals = np.repeat(n[None], len(p), axis=0)
    tang = np.concatenate([np.repeat(_norm(du)[None], len(p), axis=0), np.ones((len(p), 1))], axis=1)
    uv = np.stack([s * uv_scale[0], t * uv_scale[1]], axis=1)
    i = np.arange(nx * ny)
    x, y = i % nx, i // nx
    a = y * (nx + 1) + x
    idx = np.stack([a, a + 1, a + nx + 2, a, a + nx + 2, a + nx + 1], axis=1).reshape(-1)
    return Mesh(p, idx, normals, tang, uv)


def merge(meshes):
    pos, idx, nor, tan, uv = [], [], [], [], []
    base = 0
    for m in meshes:
        pos.append(m.positions); nor.append(m.normals); tan.append(m.tangents); uv.append(m.uv0)
        idx.append(m.indices + base)
        base += m.num_vertices
    return Mesh(np.concatenate(pos), np.concatenate(idx), np.concatenate(nor), np.concatenate(tan), np.concatenate(uv))


def box(lo, hi, seg=(1, 1, 1), uv_scale=1.0):
    """Axis-aligned box with outward normals, each face tessellated."""
    lo, hi = np.asarray(lo, dtype=np.float64), np.asarray(hi, dtype=np.float64)
    d = hi - lo
    X, Y, Z = np.array([d[0], 0, 0]), np.array([0, d[1], 0]), np.array([0, 0, d[2]])
    sx, sy, sz = seg
    faces = [
        grid(sx, sy, lo + Z, X, Y, (uv_scale, uv_scale)),                 # +z  (X x Y = +Z)
        grid(sy, sx, lo, Y, X, (uv_scale, uv_scale)),                     # -z  (Y x X = -Z)
        grid(sy, sz, lo + X, Y, Z, (uv_scale, uv_scale)),                 # +x  (Y x Z = +X)
        grid(sz, sy, lo, Z, Y, (uv_scale, uv_scale)),                     # -x
        grid(sz, sx, lo + Y, Z, X, (uv_scale, uv_scale)),                 # +y  (Z x X = +Y)
        grid(sx, sz, lo, X, Z, (uv_scale, uv_scale)),                     # -y
    ]
    return merge(faces)


def uv_sphere(nu, nv, radius=1.0, centre=(0, 0, 0)):
    u, v = np.meshgrid(np.linspace(0, 1, nu + 1), np.linspace(0, 1, nv + 1), indexing="xy")
    u, v = u.reshape(-1), v.reshape(-1)
    phi, theta = u * 2 * np.pi, v * np.pi
    n = np.stack([np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi), np.cos(theta)], axis=1)
    p = np.asarray(centre)[None] + radius * n
    t = np.stack([-np.sin(phi), np.cos(phi), np.zeros_like(phi), np.ones_like(phi)], axis=1)
    i = np.arange(nu * nv)
    x, y = i % nu, i // nu
    a = y * (nu + 1) + x
    idx = np.stack([a, a + nu + 1, a + nu + 2, a, a + nu + 2, a + 1], axis=1).reshape(-1)
    return Mesh(p, idx, n, t, np.stack([u, v], axis=1))


def icosphere(subdiv):
    t = (1.0 + 5 ** 0.5) / 2.0
    v = _norm(np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                        [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64))
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6],
                  [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10],
                  [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    for _ in range(subdiv):
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
        es = np.sort(e, axis=1)
        key = es[:, 0] * (len(v) + 1) + es[:, 1]
        uniq, inv = np.unique(key, return_inverse=True)
        first = np.zeros(len(uniq), dtype=np.int64)
        first[inv] = np.arange(len(key))
        mid = _norm((v[es[first, 0]] + v[es[first, 1]]) * 0.5)
        m = len(v) + inv.reshape(3, -1).T          # midpoint ids per face: ab, bc, ca
        v = np.concatenate([v, mid])
        a, b, c = f[:, 0], f[:, 1], f[:, 2]
        ab, bc, ca = m[:, 0], m[:, 1], m[:, 2]
        f = np.concatenate([np.stack([a, ab, ca], 1), np.stack([b, bc, ab], 1), np.stack([c, ca, bc], 1), np.stack([ab, bc, ca], 1)])
    return v, f


def sphere_tangent_frame(n):
    """Tangent along +phi, spherical uv (seam not duplicated: fine for procedural textures)."""
    phi = np.arctan2(n[:, 1], n[:, 0])
    theta = np.arccos(np.clip(n[:, 2], -1, 1))
    t = np.stack([-np.sin(phi), np.cos(phi), np.zeros_like(phi), np.ones_like(phi)], axis=1)
    uv = np.stack([phi / (2 * np.pi) + 0.5, theta / np.pi], axis=1)
    return t, uv


def vertex_normals(p, f):
    fn = np.cross(p[f[:, 1]] - p[f[:, 0]], p[f[:, 2]] - p[f[:, 0]])
    n = np.zeros_like(p)
    for k in range(3):
        np.add.at(n, f[:, k], fn)
    return _norm(n)


def capsule_tube(p0, p1, radius, nseg=8, nring=6):
    """Open-ended tube with hemispherical caps between p0 and p1 (used for the skinned figure)."""
    p0, p1 = np.asarray(p0, dtype=np.float64), np.asarray(p1, dtype=np.float64)
    axis = _norm(p1 - p0)
    helper = np.array([1.0, 0, 0]) if abs(axis[0]) < 0.9 else np.array([0, 1.0, 0])
    bx = _norm(np.cross(axis, helper))
    by = np.cross(axis, bx)
    length = np.linalg.norm(p1 - p0)
    rows = []
    for i in range(nring + 1):            # bottom cap
        a = -np.pi / 2 + (np.pi / 2) * i / nring
        rows.append((radius * np.cos(a), radius * np.sin(a), 0.0))
    for i in range(1, nring + 1):         # top cap
        a = (np.pi / 2) * i / nring
        rows.append((radius * np.cos(a), radius * np.sin(a), length))
    pos, nor, uv, tfrac = [], [], [], []
    total = len(rows)
    for r, (rr, h, base) in enumerate(rows):
        for s in range(nseg + 1):
            ang = 2 * np.pi * s / nseg
            radial = np.cos(ang) * bx + np.sin(ang) * by
            c = p0 + axis * base
            pos.append(c + radial * rr + axis * h)
            nrm = radial * rr + axis * h
            nor.append(nrm / max(np.linalg.norm(nrm), 1e-12) if np.linalg.norm(nrm) > 1e-12 else axis * np.sign(h if h != 0 else -1))
            uv.append((s / nseg, r / (total - 1)))
            tfrac.append((base + h + radius) / (length + 2 * radius))
    pos, nor, uv = np.array(pos), np.array(nor), np.array(uv)
    idx = []
    for r in range(total - 1):
        for s in range(nseg):
            a = r * (nseg + 1) + s
            idx += [a, a + 1, a + nseg + 2, a, a + nseg + 2, a + nseg + 1]
    tang = np.concatenate([np.cross(np.repeat(axis[None], len(pos), 0), nor), np.ones((len(pos), 1))], axis=1)
    bad = np.linalg.norm(tang[:, :3], axis=1) < 1e-6
    tang[bad, :3] = bx
    tang[:, :3] = _norm(tang[:, :3])
    return Mesh(pos, np.array(idx), nor, tang, uv), np.clip(np.array(tfrac), 0, 1)
