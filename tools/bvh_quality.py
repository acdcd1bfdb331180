"""CPU experiment: how much surface-area cost does the radix (Karras) split leave on the table against a surface-area split chosen
over the SAME Morton order?  Both trees keep every subtree a contiguous range of the sorted triangles, which is what the leaf
clusters and the segment-tree fit in accel.hip rely on.  Prints sum of inner-node areas / root area (the expected number of
node visits of a random ray, up to a constant) for each splitter.   usage: python tools/bvh_quality.py [sponza|helmet|test] [candidates]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from gltf_renderer_amd import scenes


def world_triangles(s):
    out = []
    for mesh, T, _ in s.mesh_records:
        p = np.asarray(mesh.positions, np.float64)
        p = p @ T[:3, :3].T + T[:3, 3]
        st = mesh.index_stream()
        idx = np.asarray(st[0], np.int64).reshape(-1, 3) if st[0] is not None else np.arange(len(p)).reshape(-1, 3)
        out.append(p[idx])
    return np.concatenate(out)


def expand21(v):
    v = v.astype(np.uint64) & np.uint64(0x1fffff)
    for sh, m in ((32, 0x1f00000000ffff), (16, 0x1f0000ff0000ff), (8, 0x100f00f00f00f00f), (4, 0x10c30c30c30c30c3), (2, 0x1249249249249249)):
        v = (v | (v << np.uint64(sh))) & np.uint64(m)
    return v


def area(lo, hi):
    d = np.maximum(hi - lo, 0)
    return d[..., 0] * d[..., 1] + d[..., 1] * d[..., 2] + d[..., 2] * d[..., 0]


def build(keys, lo, hi, mode, cand=0):
    n = len(keys)
    total = 0.0
    depth_sum = 0
    stack = [(0, n - 1, 0)]
    root_area = area(lo.min(0), hi.max(0))
    max_depth = 0
    while stack:
        a, b, d = stack.pop()
        if a == b:
            depth_sum += d; max_depth = max(max_depth, d)
            continue
        total += area(lo[a:b + 1].min(0), hi[a:b + 1].max(0))
        # radix split: last index whose key shares the longer prefix with keys[a]
        x = int(keys[a]) ^ int(keys[b])
        if x == 0:
            radix = (a + b) >> 1
        else:
            bit = x.bit_length() - 1
            radix = a + int(np.searchsorted(keys[a:b + 1] >> np.uint64(bit) & np.uint64(1), 1)) - 1
        s = radix
        if mode != "radix" and b - a >= 2:
            m = b - a + 1
            pl = np.minimum.accumulate(lo[a:b + 1], 0); ph = np.maximum.accumulate(hi[a:b + 1], 0)
            sl = np.minimum.accumulate(lo[a:b + 1][::-1], 0)[::-1]; sh = np.maximum.accumulate(hi[a:b + 1][::-1], 0)[::-1]
            k = np.arange(1, m)                                   # left takes k triangles
            cost = area(pl[:-1], ph[:-1]) * k + area(sl[1:], sh[1:]) * (m - k)
            if mode == "sweep" or m - 1 <= cand:
                pick = int(np.argmin(cost))
            else:                                                 # `cand` evenly spaced candidates + the radix split
                c = np.unique(np.concatenate([np.linspace(0, m - 2, cand).round().astype(int), [radix - a]]))
                pick = int(c[np.argmin(cost[c])])
            s = a + pick
        stack.append((a, s, d + 1)); stack.append((s + 1, b, d + 1))
    return total / root_area, depth_sum / n, max_depth


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "sponza"
    cand = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    s = {"sponza": scenes.sponza_class, "helmet": scenes.helmet_class, "test": scenes.test_scene}[which]()
    t = world_triangles(s)
    lo, hi = t.min(1), t.max(1)
    c = (lo + hi) * 0.5
    q = (c - c.min(0)) / np.maximum(c.max(0) - c.min(0), 1e-30)
    g = np.clip(q * 2097152.0, 0, 2097151).astype(np.uint64)
    keys = (expand21(g[:, 0]) << np.uint64(2)) | (expand21(g[:, 1]) << np.uint64(1)) | expand21(g[:, 2])
    order = np.argsort(keys, kind="stable")
    keys, lo, hi = keys[order], lo[order], hi[order]
    print("%s: %d triangles" % (s.name, len(keys)))
    for mode in ("radix", "sampled", "sweep"):
        r = build(keys, lo, hi, mode, cand)
        print("%-8s sum(inner area)/root area = %9.2f   mean leaf depth %.1f   max depth %d" % (mode + (str(cand) if mode == "sampled" else ""), *r))
