// accel.hip -- on-device LBVH build for gfx950 (replaces the D3D12 driver's BLAS/TLAS builds,
// Source/RayTracingAccelerationStructure.cpp:157,213,289 as driven by Pathtracer.cpp:138-257).
//
// The reference keeps one BLAS per primitive and rebuilds a TLAS of <= 1000 instances every frame.
// Here every instance is flattened into ONE world-space triangle soup and a single BVH2 is built over
// it (no per-instance ray transform, no overlapping instance boxes to enter):
//   1. setup     : (instance, primitive) -> world-space 48-B packet, centroid bounds per block, then one block folds them
//   2. morton    : 63-bit Morton code of the centroid (21 bits / axis)
//   3. sort      : stable radix sort of (code, triangle id), seven 9-bit passes         (sort_scan.hip)
//   4. hierarchy : Karras 2012 radix tree, one lane per internal node
//   5. fit       : a min/max segment tree over the sorted triangles' boxes; every radix-tree node covers a contiguous range
//                  of them, so its box is one O(log count) range query -- no inter-lane hand-over, no fences
// Every stage streams its arrays once (HBM-bound; 48 B + 8 B + 64 B per triangle).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <algorithm>
#include <vector>

#include "pt_math.h"
#include "pt_types.h"
#include "pt_host.h"

namespace pt {


__device__ __forceinline__ uint32_t float_to_sortable(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float sortable_to_float(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// Folds the per-block centroid bounds k_setup left (6 floats per block) into bounds[6] (sortable uints).  One block.
__global__ __launch_bounds__(256) void k_bounds(const float* __restrict__ partial, uint32_t n_blocks, uint32_t* __restrict__ bounds) {
    __shared__ float red[4][6];
    float v[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (uint32_t b = threadIdx.x; b < n_blocks; b += 256) {
#pragma unroll
        for (int a = 0; a < 3; a++) { v[a] = fminf(v[a], partial[b * 6 + a]); v[3 + a] = fmaxf(v[3 + a], partial[b * 6 + 3 + a]); }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int a = 0; a < 3; a++) { v[a] = fminf(v[a], __shfl_down(v[a], off, 64)); v[3 + a] = fmaxf(v[3 + a], __shfl_down(v[3 + a], off, 64)); }
    }
    if ((threadIdx.x & 63) == 0) for (int a = 0; a < 6; a++) red[threadIdx.x >> 6][a] = v[a];
    __syncthreads();
    if (threadIdx.x < 6) {
        const int a = threadIdx.x;
        float r = red[0][a];
        for (int w = 1; w < 4; w++) r = a < 3 ? fminf(r, red[w][a]) : fmaxf(r, red[w][a]);
        bounds[a] = float_to_sortable(r);
    }
}

__device__ __forceinline__ uint32_t load_index(const BufferRec* buffers, int desc, uint32_t i) {
    if (desc == -1) return i;
    const BufferRec& b = buffers[desc];
    return b.format == PT_FORMAT_R16_UINT ? (uint32_t)((const uint16_t*)b.ptr)[i] : ((const uint32_t*)b.ptr)[i];
}

// 1. one lane per triangle: find its instance (binary search on tri_offset), build the world-space packet.
__global__ __launch_bounds__(256) void k_setup(const BufferRec* __restrict__ buffers, const InstanceRec* __restrict__ instances, int n_inst,
                                               uint32_t n_tris, TriPacket* __restrict__ out, float* __restrict__ block_bounds) {
    __shared__ float red[4][6];
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    vec3 c = v3(0);
    bool valid = i < n_tris;
    if (valid) {
        int lo = 0, hi = n_inst - 1;
        while (lo < hi) {
            int mid = (lo + hi + 1) >> 1;
            if (instances[mid].tri_offset <= i) lo = mid; else hi = mid - 1;
        }
        const InstanceRec& in = instances[lo];
        uint32_t prim = i - in.tri_offset;
        const float* pos = (const float*)buffers[in.gpu.position_descriptor].ptr;
        vec3 w[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            uint32_t vi = load_index(buffers, in.gpu.index_descriptor, prim * 3 + k);
            w[k] = mul_point(in.gpu.transform, v3p(pos + (size_t)vi * 3));
        }
        TriPacket t;
        t.v0[0] = w[0].x; t.v0[1] = w[0].y; t.v0[2] = w[0].z; t.inst = (uint32_t)lo;
        vec3 e1 = w[1] - w[0], e2 = w[2] - w[0];
        t.e1[0] = e1.x; t.e1[1] = e1.y; t.e1[2] = e1.z; t.prim = prim;
        t.e2[0] = e2.x; t.e2[1] = e2.y; t.e2[2] = e2.z; t.flags = in.mask_flags;
        out[i] = t;
        vec3 mn = hmin(hmin(w[0], w[1]), w[2]), mx = hmax(hmax(w[0], w[1]), w[2]);
        c = (mn + mx) * 0.5f;
    }
    // min/max of the centroid: wave shuffle, then the block's four waves through LDS; one 24-B row per block, folded by k_bounds
    // (float atomics on six shared words serialise: 280 us for 257 k triangles, against 6 us this way)
    float mnx = valid ? c.x : INFINITY, mny = valid ? c.y : INFINITY, mnz = valid ? c.z : INFINITY;
    float mxx = valid ? c.x : -INFINITY, mxy = valid ? c.y : -INFINITY, mxz = valid ? c.z : -INFINITY;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mnx = fminf(mnx, __shfl_down(mnx, off, 64)); mny = fminf(mny, __shfl_down(mny, off, 64)); mnz = fminf(mnz, __shfl_down(mnz, off, 64));
        mxx = fmaxf(mxx, __shfl_down(mxx, off, 64)); mxy = fmaxf(mxy, __shfl_down(mxy, off, 64)); mxz = fmaxf(mxz, __shfl_down(mxz, off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        float* r = red[threadIdx.x >> 6];
        r[0] = mnx; r[1] = mny; r[2] = mnz; r[3] = mxx; r[4] = mxy; r[5] = mxz;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int a = threadIdx.x;
        float r = red[0][a];
        for (int w = 1; w < 4; w++) r = a < 3 ? fminf(r, red[w][a]) : fmaxf(r, red[w][a]);
        block_bounds[blockIdx.x * 6 + a] = r;
    }
}

__device__ __forceinline__ uint64_t expand21(uint64_t v) {
    v &= 0x1fffff;
    v = (v | v << 32) & 0x1f00000000ffffull;
    v = (v | v << 16) & 0x1f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}

// 2. Morton codes
__global__ __launch_bounds__(256) void k_morton(const TriPacket* __restrict__ tris, uint32_t n, const uint32_t* __restrict__ bounds,
                                                uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    vec3 lo = v3(sortable_to_float(bounds[0]), sortable_to_float(bounds[1]), sortable_to_float(bounds[2]));
    vec3 hi = v3(sortable_to_float(bounds[3]), sortable_to_float(bounds[4]), sortable_to_float(bounds[5]));
    vec3 ext = hmax(hi - lo, v3(1e-30f));
    const TriPacket& t = tris[i];
    vec3 a = v3p(t.v0), b = a + v3p(t.e1), c = a + v3p(t.e2);
    vec3 cen = (hmin(hmin(a, b), c) + hmax(hmax(a, b), c)) * 0.5f;
    vec3 q = (cen - lo) / ext;
    uint64_t x = (uint64_t)clampf(q.x * 2097152.f, 0.f, 2097151.f), y = (uint64_t)clampf(q.y * 2097152.f, 0.f, 2097151.f),
             z = (uint64_t)clampf(q.z * 2097152.f, 0.f, 2097151.f);
    keys[i] = expand21(x) << 2 | expand21(y) << 1 | expand21(z);
    vals[i] = i;
}

__device__ __forceinline__ int delta(const uint64_t* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    uint64_t x = keys[i] ^ keys[j];
    if (x) return __clzll((long long)x);
    return 64 + __clz(i ^ j);
}

// 4. Karras radix tree: internal node i in [0, n-2]; leaves are ~index.
__global__ __launch_bounds__(256) void k_hierarchy(const uint64_t* __restrict__ keys, int n, BvhNode* __restrict__ nodes,
                                                   int32_t* __restrict__ node_parent, int32_t* __restrict__ leaf_parent) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t <= 1) break;
    }
    int gamma = i + s * d + min(d, 0);
    int left = (min(i, j) == gamma) ? ~gamma : gamma;
    int right = (max(i, j) == gamma + 1) ? ~(gamma + 1) : (gamma + 1);
    nodes[i].child0 = left;
    nodes[i].child1 = right;
    nodes[i]._pad[0] = (uint32_t)min(i, j);                    // the sorted-triangle range this subtree covers: first, count
    nodes[i]._pad[1] = (uint32_t)(max(i, j) - min(i, j) + 1);
    if (left < 0) leaf_parent[gamma] = i * 2; else node_parent[gamma] = i * 2;
    if (right < 0) leaf_parent[gamma + 1] = i * 2 + 1; else node_parent[gamma + 1] = i * 2 + 1;
    if (i == 0) node_parent[0] = -1;
}

// 3b. gather packets into sorted order
__global__ __launch_bounds__(256) void k_reorder(const TriPacket* __restrict__ in, const uint32_t* __restrict__ vals, uint32_t n, TriPacket* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4* s = (const float4*)(in + vals[i]);
    float4* d = (float4*)(out + i);
    d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
}

// 3c. the shading packet of every (sorted) triangle: de-index the instance's streams once, here, instead of at every hit
__global__ __launch_bounds__(256) void k_shade_packets(const TriPacket* __restrict__ tris, uint32_t n, const InstanceRec* __restrict__ instances,
                                                       ShadePacket* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const InstanceRec& in = instances[tris[i].inst];
    const uint32_t prim = tris[i].prim;
    ShadePacket p;
    memset(&p, 0, sizeof(p));
    for (int k = 0; k < 3; k++) {
        uint32_t v = prim * 3 + k;
        if (in.p_index) v = in.index_is16 ? (uint32_t)((const uint16_t*)in.p_index)[v] : ((const uint32_t*)in.p_index)[v];
        ShadePacket::V& o = p.v[k];
        const float* q = in.p_position + (size_t)v * 3;
        o.pos[0] = q[0]; o.pos[1] = q[1]; o.pos[2] = q[2];
        if (in.p_tangent_space) o.tangent_space = in.p_tangent_space[v];
        if (in.p_texcoord[0]) { float2 t = in.p_texcoord[0][v]; p.uv0[k][0] = t.x; p.uv0[k][1] = t.y; }
        if (in.p_texcoord[1]) { float2 t = in.p_texcoord[1][v]; p.uv1[k][0] = t.x; p.uv1[k][1] = t.y; }
        if (in.p_color) { uint2 c = in.p_color[v]; p.color[k][0] = c.x; p.color[k][1] = c.y; }
    }
    p.inst = tris[i].inst;
    const float4* s = (const float4*)&p;
    float4* d = (float4*)(out + i);
#pragma unroll
    for (int q = 0; q < 8; q++) d[q] = s[q];
}

// 5. fit.  A radix-tree node covers a contiguous range [first, first + count) of the sorted triangles (k_hierarchy left it in
//    _pad), so its box is a range min/max over their boxes.  T is a heap-ordered segment tree over P (a power of two >= n)
//    leaf slots: T[P + i] = box of triangle i, T[k] = T[2k] U T[2k+1], root T[1].  It is built eight levels per launch through
//    LDS; a query unites O(log count) nodes.  min/max are exact, so the boxes are those of any other summation order.
//    (The textbook alternative -- lanes climb from the leaves and the second arriver at a node carries on -- needs an
//    agent-scope release/acquire pair per level, which on eight XCDs with separate L2s costs an L2 write-back and invalidate each
//    time: 700 us for 257 k triangles, against ~35 us for the tree and the queries.)
struct __attribute__((aligned(16))) SegBox { float lo[3], _a, hi[3], _b; };
static_assert(sizeof(SegBox) == 32, "SegBox");

// One pass: each block takes 256 consecutive entries of the input level (heap indices in_base + e; in_base is a power of two and
// equals that level's size) and writes the eight levels above them.  LEAVES: the input entries are the triangles' boxes, computed
// here and stored as level P; otherwise they are read back from T.  Entries >= in_valid count as empty.
template <bool LEAVES>
__global__ __launch_bounds__(256) void k_seg_pass(const TriPacket* __restrict__ tris, SegBox* __restrict__ T, uint32_t in_base, uint32_t in_valid) {
    __shared__ float sh[6][256];
    const uint32_t t = threadIdx.x, e = blockIdx.x * 256u + t;
    float b[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
    if (e < in_valid) {
        if (LEAVES) {
            const TriPacket& tp = tris[e];
            const vec3 p = v3p(tp.v0), q = p + v3p(tp.e1), r = p + v3p(tp.e2);
            // (this expression for the box is also the traversal's, pt_traverse.h candidate_stands: a hit must pass the box test of its own box)
            const vec3 lo = hmin(hmin(p, q), r), hi = hmax(hmax(p, q), r);
            b[0] = lo.x; b[1] = lo.y; b[2] = lo.z; b[3] = hi.x; b[4] = hi.y; b[5] = hi.z;
            SegBox o; o.lo[0] = b[0]; o.lo[1] = b[1]; o.lo[2] = b[2]; o._a = 0; o.hi[0] = b[3]; o.hi[1] = b[4]; o.hi[2] = b[5]; o._b = 0;
            T[in_base + e] = o;
        } else {
            const SegBox& s = T[in_base + e];
            b[0] = s.lo[0]; b[1] = s.lo[1]; b[2] = s.lo[2]; b[3] = s.hi[0]; b[4] = s.hi[1]; b[5] = s.hi[2];
        }
    }
#pragma unroll
    for (int a = 0; a < 6; a++) sh[a][t] = b[a];
    __syncthreads();
    for (int k = 1; k <= 8; k++) {
        const bool act = t < (256u >> k);
        if (act) {
#pragma unroll
            for (int a = 0; a < 3; a++) { b[a] = fminf(sh[a][2 * t], sh[a][2 * t + 1]); b[3 + a] = fmaxf(sh[3 + a][2 * t], sh[3 + a][2 * t + 1]); }
        }
        __syncthreads();
        if (act) {
#pragma unroll
            for (int a = 0; a < 6; a++) sh[a][t] = b[a];
            // entry (blockIdx*256 + (t << k)) of the input level folds into entry >> k of the level k above
            if ((in_base >> k) != 0 && blockIdx.x * 256u + (t << k) < in_base) {
                SegBox o; o.lo[0] = b[0]; o.lo[1] = b[1]; o.lo[2] = b[2]; o._a = 0; o.hi[0] = b[3]; o.hi[1] = b[4]; o.hi[2] = b[5]; o._b = 0;
                T[(in_base >> k) + ((blockIdx.x * 256u) >> k) + t] = o;
            }
        }
        __syncthreads();
    }
}

__device__ __forceinline__ void seg_query(const SegBox* __restrict__ T, uint32_t P, uint32_t first, uint32_t count, float* lo, float* hi) {
    float b[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
    auto take = [&](uint32_t k) {
        const SegBox& s = T[k];
#pragma unroll
        for (int a = 0; a < 3; a++) { b[a] = fminf(b[a], s.lo[a]); b[3 + a] = fmaxf(b[3 + a], s.hi[a]); }
    };
    for (uint32_t l = first + P, r = first + count + P; l < r; l >>= 1, r >>= 1) {
        if (l & 1u) take(l++);
        if (r & 1u) take(--r);
    }
    lo[0] = b[0]; lo[1] = b[1]; lo[2] = b[2]; hi[0] = b[3]; hi[1] = b[4]; hi[2] = b[5];
}

// The ranges of a radix-tree node's two children, from the node alone: it covers [first, first + count) and splits after gamma,
// which is the index of its left child whatever that is (leaf gamma or inner node gamma; the right child is gamma + 1).
struct ChildRanges { uint32_t f0, k0, f1, k1; };
__device__ __forceinline__ ChildRanges child_ranges(const BvhNode& n) {
    const uint32_t first = n._pad[0], count = n._pad[1];
    const uint32_t gamma = (uint32_t)(n.child0 < 0 ? ~n.child0 : n.child0);
    ChildRanges r;
    r.f0 = first; r.k0 = gamma - first + 1u; r.f1 = gamma + 1u; r.k1 = count - r.k0;
    return r;
}

// one lane per radix-tree node: the boxes of its two children
__global__ __launch_bounds__(256) void k_fit(const SegBox* __restrict__ T, uint32_t P, uint32_t n_nodes, BvhNode* __restrict__ nodes) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    BvhNode& n = nodes[i];
    const ChildRanges r = child_ranges(n);
    seg_query(T, P, r.f0, r.k0, n.lo0, n.hi0);
    seg_query(T, P, r.f1, r.k1, n.lo1, n.hi1);
}

// Writes one wide node: the children's boxes on the 8-bit grid of the node's own box (pt_types.h Bvh4Node).  Conservative: every
// lo is rounded down and every hi up, and each is checked against bvh_dequant, the expression the traversal evaluates.
__device__ void wide_write(Bvh4Node* dst, const float (*lo)[3], const float (*hi)[3], const int32_t* ref, int cnt) {
    constexpr int W = kBvhWidth, QW = W / 4;               // QW words per bound (byte k & 3 of word k >> 2 = child k)
    uint32_t qlo[3][QW], qhi[3][QW], exps = 0, origin[3];
    for (int a = 0; a < 3; a++) {
        for (int q = 0; q < QW; q++) qlo[a][q] = qhi[a][q] = 0;
        float p = lo[0][a], top = hi[0][a];
        for (int k = 1; k < cnt; k++) { p = fminf(p, lo[k][a]); top = fmaxf(top, hi[k][a]); }
        // smallest power-of-two step whose 255th plane reaches the far side
        uint32_t e = (__float_as_uint((top - p) * (1.0f / 255.0f)) >> 23) & 0xffu;
        e = e < 1u ? 1u : (e > 254u ? 254u : e);
        while (e < 254u && !(bvh_dequant(255u, bvh_step(e), p) >= top)) e++;
        const float step = bvh_step(e), inv_step = bvh_step(254u - e);
        for (int k = 0; k < cnt; k++) {
            int ql = (int)floorf((lo[k][a] - p) * inv_step), qh = (int)ceilf((hi[k][a] - p) * inv_step);
            ql = ql < 0 ? 0 : (ql > 255 ? 255 : ql); qh = qh < 0 ? 0 : (qh > 255 ? 255 : qh);
            while (ql > 0 && bvh_dequant((uint32_t)ql, step, p) > lo[k][a]) ql--;
            while (qh < 255 && bvh_dequant((uint32_t)qh, step, p) < hi[k][a]) qh++;
            qlo[a][k >> 2] |= (uint32_t)ql << (8 * (k & 3)); qhi[a][k >> 2] |= (uint32_t)qh << (8 * (k & 3));
        }
        origin[a] = __float_as_uint(p);
        exps |= e << (8 * a);
    }
    uint32_t word[kNodeFloat4 * 4];
    for (int k = 0; k < kNodeFloat4 * 4; k++) word[k] = 0;
    word[0] = origin[0]; word[1] = origin[1]; word[2] = origin[2]; word[3] = exps;
    for (int k = 0; k < W; k++) word[4 + k] = (uint32_t)(k < cnt ? ref[k] : kEmptyChild);
    int at = 4 + W;
    if (W == 4) {                                           // [qlox qhix qloy qhiy] [qloz qhiz 0 0]
        word[at + 0] = qlo[0][0]; word[at + 1] = qhi[0][0]; word[at + 2] = qlo[1][0]; word[at + 3] = qhi[1][0]; word[at + 4] = qlo[2][0]; word[at + 5] = qhi[2][0];
    } else {                                                // one dwordx4 per axis: [qlo 0-3, qlo 4-7, qhi 0-3, qhi 4-7]
        for (int a = 0; a < 3; a++) { word[at + 4 * a + 0] = qlo[a][0]; word[at + 4 * a + 1] = qlo[a][QW - 1]; word[at + 4 * a + 2] = qhi[a][0]; word[at + 4 * a + 3] = qhi[a][QW - 1]; }
    }
    uint4* d = (uint4*)dst;
#pragma unroll
    for (int q = 0; q < kNodeFloat4; q++) d[q] = make_uint4(word[4 * q], word[4 * q + 1], word[4 * q + 2], word[4 * q + 3]);
}

// 6. collapse to wide nodes (kBvhWidth children), greedily by surface area, level by level of the WIDE tree, top down.  A frontier entry is a binary node
//    that becomes a wide node; it starts with its two children and keeps opening the inner child with the largest box until it
//    holds kBvhWidth children or only leaves.  Compared with "keep every even level of the binary tree" this fills the slots (about
//    3.0 -> 3.6 children per node on the Sponza-class scene), so the tree is shallower and a ray visits fewer nodes.
//    counters[0]: wide nodes allocated so far; counters[1 + L]: size of level L's frontier.
#ifndef PT_CHILD_ORDER
#define PT_CHILD_ORDER 1      // children stored in decreasing surface area (measured 0 / 1 / 2 = as collapsed / by area / by triangle count, Mrays/s at 8 spp,
                            // Sponza class: 4813 / 4906 / 4881; material grid 3002 / 3006 / 2982)
#endif
constexpr int kCollapseMaxLevels = 4096;
constexpr int kCollapseNeedSlot = kCollapseMaxLevels + 8;       // counters[] slot: the deepest traversal stack any ray can need (entries)
__device__ void collapse_one(const BvhNode* __restrict__ nodes2, uint32_t binary_node, uint32_t wide_index, uint32_t level,
                             uint32_t* __restrict__ frontier_out, uint32_t* __restrict__ widx_out, uint32_t* __restrict__ counters,
                             Bvh4Node* __restrict__ out, WideRanges* __restrict__ ranges_out, uint32_t need_in, uint32_t* __restrict__ need_out) {
    // References while collapsing: a binary node index (>= 0) or a leaf reference (< 0).  An inner subtree of at most kLeafMax
    // triangles becomes ONE leaf (its triangles are contiguous): the bottom of a binary tree is full of 2- and 3-triangle
    // subtrees, which as wide nodes would spend a whole node step on two boxes.  Its size follows from the parent's range.
    auto child_ref = [](int32_t child, uint32_t first, uint32_t count) -> int32_t {
        if (child < 0) return child;
        return count <= (uint32_t)kLeafMax ? ~(int32_t)(first | ((count - 1u) << 28)) : child;
    };
    const BvhNode& n = nodes2[binary_node];
    const ChildRanges nr = child_ranges(n);
    int32_t ref[kBvhWidth];
    WideRanges wr;                                          // the sorted-triangle range under every child: what a refit re-queries
    for (int k = 0; k < kBvhWidth; k++) { wr.first[k] = 0; wr.count[k] = 0; }
    for (int k = 2; k < kBvhWidth; k++) ref[k] = kEmptyChild;
    ref[0] = child_ref(n.child0, nr.f0, nr.k0); ref[1] = child_ref(n.child1, nr.f1, nr.k1);
    wr.first[0] = nr.f0; wr.count[0] = nr.k0; wr.first[1] = nr.f1; wr.count[1] = nr.k1;
    float lo[kBvhWidth][3], hi[kBvhWidth][3];
    for (int a = 0; a < 3; a++) { lo[0][a] = n.lo0[a]; hi[0][a] = n.hi0[a]; lo[1][a] = n.lo1[a]; hi[1][a] = n.hi1[a]; }
    int cnt = 2;
    while (cnt < kBvhWidth) {
        int pick = -1;
        float best = -1.0f;
        for (int k = 0; k < cnt; k++) {
            if (ref[k] < 0) continue;
            const float dx = hi[k][0] - lo[k][0], dy = hi[k][1] - lo[k][1], dz = hi[k][2] - lo[k][2];
            const float area = dx * dy + dy * dz + dz * dx;
            if (area > best) { best = area; pick = k; }
        }
        if (pick < 0) break;
        const BvhNode& m = nodes2[ref[pick]];
        const ChildRanges mr = child_ranges(m);
        ref[pick] = child_ref(m.child0, mr.f0, mr.k0); ref[cnt] = child_ref(m.child1, mr.f1, mr.k1);
#pragma unroll
        for (int k = 0; k < kBvhWidth; k++) {                  // (static indices: a dynamically indexed local array lives in scratch)
            if (k == pick) { wr.first[k] = mr.f0; wr.count[k] = mr.k0; }
            if (k == cnt) { wr.first[k] = mr.f1; wr.count[k] = mr.k1; }
        }
        for (int a = 0; a < 3; a++) { lo[pick][a] = m.lo0[a]; hi[pick][a] = m.hi0[a]; lo[cnt][a] = m.lo1[a]; hi[cnt][a] = m.hi1[a]; }
        cnt++;
    }
    // children that stay inner nodes become wide nodes of the next level: one pair of atomics per wave reserves their frontier
    // slots and wide-node indices (the two advance together)
    uint32_t mine = 0;
    for (int k = 0; k < cnt; k++) mine += ref[k] >= 0 ? 1u : 0u;
    uint32_t incl = mine;
    const uint32_t lane = __lane_id();
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off, 64);
        if (lane >= (uint32_t)off) incl += up;
    }
    const unsigned long long active = __ballot(1);
    const int last = 63 - __clzll((long long)active), leader = __ffsll((long long)active) - 1;
    const uint32_t total = __shfl(incl, last, 64);
    uint32_t base_f = 0, base_w = 0;
    if ((int)lane == leader && total) { base_f = atomicAdd(counters + 2 + level, total); base_w = atomicAdd(counters + 0, total); }
    base_f = __shfl(base_f, leader, 64); base_w = __shfl(base_w, leader, 64);
    // Traversal-stack entries a ray can hold below this node: what it held on arrival plus the cnt - 1 siblings it pushes here.
    // The maximum over all nodes is the stack a ray can ever need; one atomicMax per wave (running maximum up the active lanes,
    // which are a prefix of the wave).
    const uint32_t need = need_in + (uint32_t)cnt - 1u;
    uint32_t run = need;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(run, off, 64);
        if (lane >= (uint32_t)off) run = max(run, up);
    }
    if ((int)lane == last) atomicMax(counters + kCollapseNeedSlot, run);
    uint32_t slot = incl - mine;
    for (int k = 0; k < cnt; k++) {
        if (ref[k] >= 0) {
            frontier_out[base_f + slot] = (uint32_t)ref[k];
            widx_out[base_f + slot] = base_w + slot;
            need_out[base_f + slot] = need;
            ref[k] = (int32_t)(base_w + slot);
            slot++;
        }
    }
#if PT_CHILD_ORDER
    // Children in decreasing surface area (PT_CHILD_ORDER 1) / triangle count (2): occlusion rays enter the children in slot order
    // (any hit ends them, so they skip the distance sort) and the biggest child is the likeliest to hold an occluder.  Closest-hit
    // rays sort by distance anyway.  A 4-element sorting network on static indices (dynamic indexing would put the arrays in scratch).
    {
        float key[kBvhWidth];
#pragma unroll
        for (int k = 0; k < kBvhWidth; k++) {
            const float dx = hi[k][0] - lo[k][0], dy = hi[k][1] - lo[k][1], dz = hi[k][2] - lo[k][2];
            key[k] = k < cnt ? (PT_CHILD_ORDER == 2 ? (float)wr.count[k] : dx * dy + dy * dz + dz * dx) : -1.0f;
        }
#define PT_CSWAP_CHILD(A, B)                                                                                              \
        if (key[A] < key[B]) {                                                                                             \
            float tk = key[A]; key[A] = key[B]; key[B] = tk;                                                               \
            int32_t tr = ref[A]; ref[A] = ref[B]; ref[B] = tr;                                                             \
            uint32_t tf = wr.first[A]; wr.first[A] = wr.first[B]; wr.first[B] = tf;                                       \
            uint32_t tc = wr.count[A]; wr.count[A] = wr.count[B]; wr.count[B] = tc;                                       \
            for (int a = 0; a < 3; a++) { float t0 = lo[A][a]; lo[A][a] = lo[B][a]; lo[B][a] = t0; float t1 = hi[A][a]; hi[A][a] = hi[B][a]; hi[B][a] = t1; } \
        }
        static_assert(kBvhWidth == 4 || !PT_CHILD_ORDER, "child ordering is written for 4-wide nodes");
        PT_CSWAP_CHILD(0, 1) PT_CSWAP_CHILD(2, 3) PT_CSWAP_CHILD(0, 2) PT_CSWAP_CHILD(1, 3) PT_CSWAP_CHILD(1, 2)
#undef PT_CSWAP_CHILD
    }
#endif
    wide_write(out + wide_index, lo, hi, ref, cnt);
    ranges_out[wide_index] = wr;
}

__global__ __launch_bounds__(256) void k_collapse_init(uint32_t* __restrict__ frontier, uint32_t* __restrict__ widx, uint32_t* __restrict__ need,
                                                       uint32_t* __restrict__ counters) {
    for (int k = threadIdx.x; k < kCollapseMaxLevels + 16; k += 256) counters[k] = k <= 1 ? 1u : 0u;   // one wide node (the root), level 0 holds one entry
    if (threadIdx.x == 0) { frontier[0] = 0u; widx[0] = 0u; need[0] = 0u; }                         // binary node 0 -> wide node 0
}

// One level per launch (large scenes).  The host launches several levels without looking, with grids sized for the largest
// frontier the level can have, so a level may well be empty.
__global__ __launch_bounds__(256) void k_collapse_level(const BvhNode* __restrict__ nodes2, const uint32_t* __restrict__ frontier_in,
                                                        const uint32_t* __restrict__ widx_in, uint32_t level, uint32_t* __restrict__ frontier_out,
                                                        uint32_t* __restrict__ widx_out, uint32_t* __restrict__ counters, Bvh4Node* __restrict__ out,
                                                        WideRanges* __restrict__ ranges_out, const uint32_t* __restrict__ need_in, uint32_t* __restrict__ need_out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= counters[1 + level]) return;
    collapse_one(nodes2, frontier_in[i], widx_in[i], level, frontier_out, widx_out, counters, out, ranges_out, need_in[i], need_out);
}

// All levels in one launch of ONE workgroup (small scenes: a rebuild per animation frame is launch-bound, and sixteen level
// launches were 0.22 of its 0.39 ms).  Levels are separated by workgroup barriers; the frontier ping-pongs between fr[0] and fr[1].
constexpr uint32_t kSmallCollapseNodes = 32768;
__global__ __launch_bounds__(1024) void k_collapse_small(const BvhNode* __restrict__ nodes2, uint32_t* fr0, uint32_t* wi0, uint32_t* fr1, uint32_t* wi1,
                                                         uint32_t* counters, Bvh4Node* __restrict__ out, WideRanges* __restrict__ ranges_out, uint32_t* nd0, uint32_t* nd1) {
    for (uint32_t level = 0; level < (uint32_t)kCollapseMaxLevels; level++) {
        // (the counters are advanced by agent-scope atomics, which are performed in L2 and leave this CU's L1 copy of the line stale)
        const uint32_t n_in = __hip_atomic_load(counters + 1 + level, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (n_in == 0) break;                                        // the same for every lane: no barrier is skipped by some
        uint32_t* fin = (level & 1u) ? fr1 : fr0;  uint32_t* win = (level & 1u) ? wi1 : wi0;
        uint32_t* fout = (level & 1u) ? fr0 : fr1; uint32_t* wout = (level & 1u) ? wi0 : wi1;
        const uint32_t* nin = (level & 1u) ? nd1 : nd0; uint32_t* nout = (level & 1u) ? nd0 : nd1;
        for (uint32_t i = threadIdx.x; i < n_in; i += blockDim.x) collapse_one(nodes2, fin[i], win[i], level, fout, wout, counters, out, ranges_out, nin[i], nout);
        __threadfence_block();
        __syncthreads();
    }
}


// ---- 4b. PLOC: parallel locally-ordered clustering (Meister & Bittner 2018) as the topology builder instead of the radix tree.
// The radix tree splits at the highest differing Morton bit -- a spatial median whatever the geometry looks like; measured on the
// Sponza-class scene its inner-node area (the expected node visits of a ray) is 58 root areas against 47 for a binned-SAH tree.
// PLOC builds bottom-up from the same Morton order: every cluster looks at its 2 * kPlocRadius neighbours in the current cluster
// array for the partner with the smallest joint surface area; pairs that chose each other merge; the survivors are compacted and
// the rounds repeat until one cluster is left.  Every later step here (segment-tree fit, leaf clusters, collapse, refit) needs each
// subtree to cover a CONTIGUOUS range of the triangle array, which clustering does not preserve, so the tree is laid out afterwards:
// a top-down pass gives every node its range [first, first + count) in depth-first leaf order and the Karras-style index the other
// kernels assume (a left child is named by the LAST leaf of its range, a right child by the FIRST of its own -- unique per inner
// node, root = 0), and the triangle packets are gathered into that order.
#ifndef PT_PLOC_RADIUS
#define PT_PLOC_RADIUS 16       // neighbours looked at on either side (measured 4 / 8 / 16 / 32 with the integer ranking below, Mrays/s at 8 spp on the
                                // Sponza-class scene / the material grid: 4804 / 2916, 4761 / 2999, 4878 / 3002, 4705 / 2968)
#endif
// (Order among partners of equal joint area, same two scenes, measured with an earlier float-comparison ranking: index distance then parity
//  4750 / 3030; parity first 4739 / 3027; farthest first 4856 / 2923 and lower index only 4840 / 3021, both unsafe -- a strip of equal
//  quads merges one pair per round; the more compact union first, then index distance and parity: what k_ploc_nearest does.)
#ifndef PT_PLOC_RADIUS_TOP
#define PT_PLOC_RADIUS_TOP 16   // ... and once fewer than kPlocTopClusters clusters are left (the upper levels; a wider search there -- 32 / 128 / 512 -- was
                                // measured and does not pay: greedy agglomeration does not profit from seeing further)
#endif
constexpr int kPlocRadius = PT_PLOC_RADIUS, kPlocRadiusTop = PT_PLOC_RADIUS_TOP, kPlocRadiusMax = PT_PLOC_RADIUS > PT_PLOC_RADIUS_TOP ? PT_PLOC_RADIUS : PT_PLOC_RADIUS_TOP;
constexpr uint32_t kPlocTopClusters = 16384;
struct PlocCluster { float lo[3]; int32_t ref; float hi[3]; uint32_t count; };     // 32 B; ref < 0: leaf ~(Morton index), >= 0: temporary node id
static_assert(sizeof(PlocCluster) == 32, "PlocCluster");
// counters: [0] temporary nodes made, [4] / [5] cluster count of even / odd rounds, [6] reinsertion moves, [7] layout error flag, [8 + L] layout frontier size of level L
constexpr int kPlocLevelBase = 8, kPlocLayoutError = 7;

__global__ __launch_bounds__(256) void k_ploc_init(const TriPacket* __restrict__ tris, uint32_t n, PlocCluster* __restrict__ c, uint32_t* __restrict__ counters,
                                                   SegBox* __restrict__ t_box) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { counters[0] = 0; counters[4] = n; counters[5] = n; }
    if (i >= n) return;
    const TriPacket& tp = tris[i];
    const vec3 p = v3p(tp.v0), q = p + v3p(tp.e1), r = p + v3p(tp.e2);
    const vec3 lo = hmin(hmin(p, q), r), hi = hmax(hmax(p, q), r);
    PlocCluster o;
    o.lo[0] = lo.x; o.lo[1] = lo.y; o.lo[2] = lo.z; o.ref = ~(int32_t)i;
    o.hi[0] = hi.x; o.hi[1] = hi.y; o.hi[2] = hi.z; o.count = 1;
    c[i] = o;
    SegBox b;                                               // the reinsertion pass (4c) keeps every node's box: slots [0, n) are the leaves
    b.lo[0] = lo.x; b.lo[1] = lo.y; b.lo[2] = lo.z; b._a = 0; b.hi[0] = hi.x; b.hi[1] = hi.y; b.hi[2] = hi.z; b._b = 0;
    t_box[i] = b;
}

// the partner with the smallest joint surface area among the 2 * radius neighbours.  A pair is ranked by an INTEGER key that is the
// same from both of its ends -- (area bits, extent-sum bits | index distance, parity of the lower index, lower index) -- a strict
// total order on pairs, so the globally best pair always chooses each other and every round merges at least one pair.  The order
// among equal areas (regular grids are full of them) matters twice: for the tree -- the more compact union first: a 2 x 2 block of
// quads and a 4 x 1 strip have the same area, the block is the better node -- and for the number of rounds -- index distance, then
// the parity of the lower index, make a run of equal boxes pair up 0-1, 2-3, 4-5 ... in ONE round, where "lower index first" chains
// every cluster to its left neighbour and merges one pair per round.  (The ranking was first written as chained float comparisons
// `a < best || (a == best && ...)`; that version stalled on a real scene -- clusters choosing i - 3 over an i - 1 with the
// bit-identical union box, nobody chosen back -- and converged again when unrelated stores were added to the loop: the integer
// key leaves the compiler nothing to reorder.)
__global__ __launch_bounds__(256) void k_ploc_nearest(const PlocCluster* __restrict__ c, const uint32_t* __restrict__ n_ptr, uint32_t* __restrict__ nn) {
    __shared__ float s_lo[3][256 + 2 * kPlocRadiusMax], s_hi[3][256 + 2 * kPlocRadiusMax];
    const uint32_t n = *n_ptr;
    const int radius = n < kPlocTopClusters ? kPlocRadiusTop : kPlocRadius;
    const int base = (int)(blockIdx.x * 256u) - radius;
    if ((uint32_t)(blockIdx.x * 256u) >= n) return;
    for (int k = threadIdx.x; k < 256 + 2 * radius; k += 256) {
        const int g = base + k;
        const bool in = g >= 0 && (uint32_t)g < n;
        const PlocCluster& q = c[in ? g : 0];
        s_lo[0][k] = in ? q.lo[0] : 0.f; s_lo[1][k] = in ? q.lo[1] : 0.f; s_lo[2][k] = in ? q.lo[2] : 0.f;
        s_hi[0][k] = in ? q.hi[0] : 0.f; s_hi[1][k] = in ? q.hi[1] : 0.f; s_hi[2][k] = in ? q.hi[2] : 0.f;
    }
    __syncthreads();
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const int me = (int)threadIdx.x + radius;
    const float lx = s_lo[0][me], ly = s_lo[1][me], lz = s_lo[2][me], hx = s_hi[0][me], hy = s_hi[1][me], hz = s_hi[2][me];
    unsigned long long best_hi = ~0ull, best_lo = ~0ull;
    uint32_t best_j = i;
    for (int d = -radius; d <= radius; d++) {
        const int g = (int)i + d;
        if (d == 0 || g < 0 || (uint32_t)g >= n) continue;
        const int k = me + d;
        const float dx = fmaxf(hx, s_hi[0][k]) - fminf(lx, s_lo[0][k]), dy = fmaxf(hy, s_hi[1][k]) - fminf(ly, s_lo[1][k]), dz = fmaxf(hz, s_hi[2][k]) - fminf(lz, s_lo[2][k]);
        float a = __fadd_rn(__fadd_rn(__fmul_rn(dx, dy), __fmul_rn(dy, dz)), __fmul_rn(dz, dx));      // explicitly rounded: the same bits from both ends
        float e = __fadd_rn(__fadd_rn(dx, dy), dz);
        // NaN, negative or overflowing values (garbage vertices) rank last but still pair up, by index: the rounds always end
        a = (a >= 0.0f) ? fminf(a, 3.0e38f) : 3.0e38f;
        e = (e >= 0.0f) ? fminf(e, 3.0e38f) : 3.0e38f;
        const uint32_t dist = (uint32_t)(d < 0 ? -d : d), m = min(i, (uint32_t)g);
        const unsigned long long key_hi = ((unsigned long long)__float_as_uint(a) << 32) | (unsigned long long)__float_as_uint(e);   // non-negative floats order like their bits
        const unsigned long long key_lo = ((unsigned long long)dist << 33) | ((unsigned long long)(m & 1u) << 32) | (unsigned long long)m;
        if (key_hi < best_hi || (key_hi == best_hi && key_lo < best_lo)) { best_hi = key_hi; best_lo = key_lo; best_j = (uint32_t)g; }
    }
    nn[i] = best_j;
}

// mutual pairs merge into the lower slot (a new temporary node); the upper slot is dropped by the compaction
__global__ __launch_bounds__(256) void k_ploc_merge(PlocCluster* __restrict__ c, const uint32_t* __restrict__ n_ptr, const uint32_t* __restrict__ nn,
                                                    uint32_t* __restrict__ valid, int32_t* __restrict__ t_left, int32_t* __restrict__ t_right,
                                                    uint32_t* __restrict__ t_count, uint32_t* __restrict__ counters, SegBox* __restrict__ t_box, uint32_t n_leaves) {
    const uint32_t n = *n_ptr;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t j = nn[i];
    const bool mutual = j != i && nn[j] == i;
    if (!mutual) { valid[i] = 1u; return; }
    if (i > j) { valid[i] = 0u; return; }
    const PlocCluster a = c[i], b = c[j];
    const uint32_t idx = atomicAdd(counters + 0, 1u);       // (the numbering is arbitrary: the layout pass renames every node)
    t_left[idx] = a.ref; t_right[idx] = b.ref; t_count[idx] = a.count + b.count;
    PlocCluster o;
    for (int k = 0; k < 3; k++) { o.lo[k] = fminf(a.lo[k], b.lo[k]); o.hi[k] = fmaxf(a.hi[k], b.hi[k]); }
    o.ref = (int32_t)idx; o.count = a.count + b.count;
    c[i] = o;
    valid[i] = 1u;
    SegBox sb;                                              // slots [n_leaves, 2 n_leaves - 1): the temporary inner nodes
    sb.lo[0] = o.lo[0]; sb.lo[1] = o.lo[1]; sb.lo[2] = o.lo[2]; sb._a = 0; sb.hi[0] = o.hi[0]; sb.hi[1] = o.hi[1]; sb.hi[2] = o.hi[2]; sb._b = 0;
    t_box[n_leaves + idx] = sb;
}

__global__ __launch_bounds__(256) void k_ploc_compact(const PlocCluster* __restrict__ in, const uint32_t* __restrict__ n_ptr, const uint32_t* __restrict__ valid,
                                                      const uint32_t* __restrict__ pos, PlocCluster* __restrict__ out, uint32_t* __restrict__ n_out) {
    const uint32_t n = *n_ptr;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (valid[i]) {
        const float4* s4 = (const float4*)(in + i);
        float4* d4 = (float4*)(out + pos[i]);
        d4[0] = s4[0]; d4[1] = s4[1];
    }
    if (i == n - 1) *n_out = pos[i] + valid[i];
}

// ---- 4c. Parallel reinsertion (Meister & Bittner 2018, "Parallel Reinsertion for Bounding Volume Hierarchy Optimization") over the
// clustering's binary tree, before it is laid out.  Greedy agglomeration decides every pair with what it sees in a window of the
// Morton order; a subtree that ended up under the wrong parent stays there.  A pass lets EVERY node look for the place in the tree
// where it would cost the least surface area: taking node `in` out removes its parent and shrinks the ancestors; putting it next
// to a node `out` adds a parent for the two and grows out's ancestors.  The search is a branch-and-bound walk without a stack: up
// the path to the root one ancestor (`pivot`) at a time, and down into the subtree hanging off each pivot for as long as the gain
// so far, less the area of `in` itself, can still beat the best found.  Moves are then applied in parallel where they do not
// touch: a move owns the tree path in -> top <- out (top = the lowest common ancestor, whose box the move leaves alone, or the
// grandparent of `in` if that is higher: its child link changes); 64-bit atomicMax of
// (gain bits, node) on every node of the path, highest gain wins, and a move is carried out only if it still holds its whole path.
// Disjoint paths also rule out a cycle (two subtrees moved into each other), and they make the winners independent: every box and
// triangle count a move changes lies on its own path, so the winner refits them itself and no tree-wide refit is needed.
// Node slots: [0, n) leaves in Morton order, n + t the temporary inner node t.  Measured (tools/bvh_reinsert_probe.cpp, the same
// algorithm on the CPU; inner-node area in root areas): Sponza class 61.4 -> 55.0 after 10 passes (binned SAH: 53.2), helmet
// class 41.7 -> 38.8, material grid 16.7 -> 15.9.
constexpr int kReinsMaxSteps = 8192;                         // bound on one node's search (a guard; the pruning ends a search long before)
PT_DEV uint32_t ploc_slot(int32_t ref, uint32_t n) { return ref < 0 ? (uint32_t)~ref : n + (uint32_t)ref; }
PT_DEV float box_area(const float4 lo, const float4 hi) {
    const float dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dy), __fmul_rn(dy, dz)), __fmul_rn(dz, dx));
}
PT_DEV float4 f4min(const float4 a, const float4 b) { return make_float4(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z), 0.f); }
PT_DEV float4 f4max(const float4 a, const float4 b) { return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), 0.f); }

// parent links (temporary inner index, -1 for the root) of every slot
__global__ __launch_bounds__(256) void k_reins_parents(const int32_t* __restrict__ t_left, const int32_t* __restrict__ t_right, uint32_t n, const PlocCluster* __restrict__ root,
                                                       int32_t* __restrict__ parent) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) parent[ploc_slot(root->ref, n)] = -1;
    if (i >= n - 1) return;
    parent[ploc_slot(t_left[i], n)] = (int32_t)i;
    parent[ploc_slot(t_right[i], n)] = (int32_t)i;
}

__global__ __launch_bounds__(256) void k_reins_find(const float4* __restrict__ box, const int32_t* __restrict__ parent, const int32_t* __restrict__ t_left,
                                                    const int32_t* __restrict__ t_right, uint32_t n, float min_gain, float* __restrict__ gain,
                                                    uint32_t* __restrict__ best_out, int32_t* __restrict__ best_top) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= 2 * n - 1) return;
    float best = 0.0f;
    uint32_t b_out = 0;
    int32_t b_top = -1;
    const int32_t p = parent[u];
    if (p >= 0 && parent[n + p] >= 0) {                          // neither the root nor a child of the root moves
        const float4 ilo = box[2 * u], ihi = box[2 * u + 1];
        const float area_in = box_area(ilo, ihi);
        float d_path = box_area(box[2 * (n + p)], box[2 * (n + p) + 1]);      // the parent goes away
        float d = d_path;
        // a move has to gain a fraction of the parent's area: the running sums carry rounding error of that scale, and a move made
        // for a gain of rounding error is undone by the next pass (and holds its path against real moves meanwhile)
        best = min_gain * d_path;
        if (!(best >= 0.0f)) best = 0.0f;
        int32_t pivot = p;
        float4 plo = ilo, phi = ihi;                             // box of `pivot` without `in` (set when the search leaves pivot's subtree)
        uint32_t out;
        { const uint32_t l = ploc_slot(t_left[p], n); out = l == u ? ploc_slot(t_right[p], n) : l; }
        bool down = true;
        for (int step = 0; step < kReinsMaxSteps; step++) {
            if (down) {
                const float4 olo = box[2 * out], ohi = box[2 * out + 1];
                const float merged = box_area(f4min(olo, ilo), f4max(ohi, ihi));
                const float here = d - merged;
                // (a move inside the parent's own subtree still re-links the grandparent: the owned path always reaches it)
                if (here > best) { best = here; b_out = out; b_top = pivot == p ? parent[n + p] : pivot; }
                const float grow = merged - box_area(olo, ohi);
                // (written so that a NaN -- garbage vertices -- ends the descent)
                if (out >= n && d - grow - area_in > best) { d -= grow; out = ploc_slot(t_left[out - n], n); }
                else down = false;
            } else {
                const int32_t po = parent[out];
                if (po == pivot) {
                    // the subtree hanging off `pivot` is done: one ancestor up
                    const float4 olo = box[2 * out], ohi = box[2 * out + 1];
                    if (pivot == p) { plo = olo; phi = ohi; } else { plo = f4min(plo, olo); phi = f4max(phi, ohi); }
                    const int32_t up = parent[n + pivot];
                    if (up < 0) break;
                    if (pivot != p) {
                        const float a_pivot = box_area(box[2 * (n + pivot)], box[2 * (n + pivot) + 1]);
                        d_path += a_pivot - box_area(plo, phi);                 // this ancestor shrinks
                        const float here = d_path - a_pivot;                     // `in` as the sibling of the shrunk ancestor itself
                        if (here > best) { best = here; b_out = n + (uint32_t)pivot; b_top = up; }
                    }
                    const uint32_t l = ploc_slot(t_left[up], n);
                    out = l == n + (uint32_t)pivot ? ploc_slot(t_right[up], n) : l;
                    pivot = up; d = d_path; down = true;
                } else if (ploc_slot(t_left[po], n) == out) { out = ploc_slot(t_right[po], n); down = true; }
                else {
                    const float4 olo = box[2 * (n + po)], ohi = box[2 * (n + po) + 1];
                    d += box_area(f4min(olo, ilo), f4max(ohi, ihi)) - box_area(olo, ohi);      // undo the growth charged on the way down
                    out = n + (uint32_t)po;
                }
            }
        }
    }
    gain[u] = b_top >= 0 ? best : 0.0f;
    best_out[u] = b_out;
    best_top[u] = b_top;
}

// One locking round.  LOCK = true: every candidate still in the race drops out if its path touches a path already won, else stamps
// max(key) on in -> top <- out.  LOCK = false: a candidate that still holds every node of its path has won: it marks the path as
// owned.  One round alone loses most moves to CHAINS -- A beats B on one node, B beats C on another, C is out although B never
// moves -- (CPU model of this scheme, Sponza class, 8 passes: inner area 61.4 -> 57.3 with one round, 55.4 with two, 55.3 with
// three or with sequential greedy selection), so a pass runs kReinsLockRounds of them.
constexpr int kReinsLockRounds = 3;
enum : uint8_t { REINS_CANDIDATE = 0, REINS_WON = 1, REINS_OUT = 2 };
template <bool LOCK>
__global__ __launch_bounds__(256) void k_reins_lock(const float* __restrict__ gain, const uint32_t* __restrict__ best_out, const int32_t* __restrict__ best_top,
                                                    const int32_t* __restrict__ parent, uint32_t n, unsigned long long* __restrict__ lock, uint8_t* __restrict__ owned,
                                                    uint8_t* __restrict__ state) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= 2 * n - 1) return;
    const float g = gain[u];
    if (!(g > 0.0f) || state[u] != REINS_CANDIDATE) return;
    const unsigned long long key = ((unsigned long long)__float_as_uint(g) << 32) | (unsigned long long)u;
    const uint32_t top = n + (uint32_t)best_top[u];
    // walk: 0 (LOCK only) is any node owned?  1 stamp / compare;  2 (verify only) mark as owned
    for (int walk = LOCK ? 0 : 1; walk < (LOCK ? 2 : 3); walk++) {
        bool ok = true;
        for (int side = 0; side < 2 && ok; side++) {
            uint32_t v = side == 0 ? u : best_out[u];
            for (int k = 0; k < 256 && v != top; k++) {
                if (walk == 0) ok = ok && !owned[v];
                else if (walk == 2) owned[v] = 1;
                else if (LOCK) atomicMax(lock + v, key);
                else ok = ok && lock[v] == key;
                const int32_t pv = parent[v];
                if (pv < 0) { ok = false; break; }               // cannot happen: top is an ancestor of both ends
                v = n + (uint32_t)pv;
            }
            if (v != top) ok = false;
        }
        if (walk == 0) ok = ok && !owned[top];
        else if (walk == 2) owned[top] = 1;
        else if (LOCK) atomicMax(lock + top, key);
        else ok = ok && lock[top] == key;
        if (walk == 0 && !ok) { state[u] = REINS_OUT; return; }
        if (walk == 1 && !LOCK) { if (!ok) return; state[u] = REINS_WON; }
    }
}

PT_DEV void reins_refit_node(float4* __restrict__ box, const int32_t* __restrict__ t_left, const int32_t* __restrict__ t_right, uint32_t* __restrict__ t_count,
                             uint32_t n, int32_t v) {
    const int32_t l = t_left[v], r = t_right[v];
    const uint32_t ls = ploc_slot(l, n), rs = ploc_slot(r, n);
    box[2 * (n + v)] = f4min(box[2 * ls], box[2 * rs]);
    box[2 * (n + v) + 1] = f4max(box[2 * ls + 1], box[2 * rs + 1]);
    t_count[v] = (l < 0 ? 1u : t_count[l]) + (r < 0 ? 1u : t_count[r]);
}

// carry out the surviving moves: their paths are disjoint, so each thread edits and refits only nodes it owns
__global__ __launch_bounds__(256) void k_reins_apply(const uint8_t* __restrict__ state, const uint32_t* __restrict__ best_out, const int32_t* __restrict__ best_top,
                                                     int32_t* __restrict__ parent, int32_t* __restrict__ t_left, int32_t* __restrict__ t_right,
                                                     uint32_t* __restrict__ t_count, float4* __restrict__ box, uint32_t n, uint32_t* __restrict__ applied) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= 2 * n - 1 || state[u] != REINS_WON) return;
    const int32_t u_ref = u < n ? ~(int32_t)u : (int32_t)(u - n);
    const int32_t p = parent[u], g = parent[n + p], top = best_top[u];
    const int32_t s_ref = t_left[p] == u_ref ? t_right[p] : t_left[p];
    // out: the sibling takes the parent's place
    if (t_left[g] == p) t_left[g] = s_ref; else t_right[g] = s_ref;
    parent[ploc_slot(s_ref, n)] = g;
    // in: the freed parent node goes between `out` and its parent
    const uint32_t o = best_out[u];
    const int32_t o_ref = o < n ? ~(int32_t)o : (int32_t)(o - n);
    const int32_t po = parent[o];
    if (t_left[po] == o_ref) t_left[po] = p; else t_right[po] = p;
    parent[n + p] = po;
    t_left[p] = o_ref; t_right[p] = u_ref;
    parent[o] = p;
    for (int side = 0; side < 2; side++) {
        int32_t v = side == 0 ? g : p;
        for (int k = 0; k < 256 && v >= 0 && v != top; k++) { reins_refit_node(box, t_left, t_right, t_count, n, v); v = parent[n + v]; }
    }
    atomicAdd(applied, 1u);
}

// layout, one level per launch: frontier entry = (temporary node, first leaf position, final index)
__global__ __launch_bounds__(256) void k_ploc_layout_init(const PlocCluster* __restrict__ c, uint32_t* __restrict__ f_node, uint32_t* __restrict__ f_first,
                                                          uint32_t* __restrict__ f_index, uint32_t* __restrict__ counters) {
    for (int k = threadIdx.x; k < kCollapseMaxLevels; k += 256) counters[kPlocLevelBase + k] = k == 0 ? 1u : 0u;
    if (threadIdx.x == 0) { f_node[0] = (uint32_t)c[0].ref; f_first[0] = 0u; f_index[0] = 0u; counters[kPlocLayoutError] = 0u; }
}
__global__ __launch_bounds__(256) void k_ploc_layout_level(const int32_t* __restrict__ t_left, const int32_t* __restrict__ t_right, const uint32_t* __restrict__ t_count,
                                                           const uint32_t* __restrict__ f_node, const uint32_t* __restrict__ f_first, const uint32_t* __restrict__ f_index,
                                                           uint32_t level, uint32_t* __restrict__ o_node, uint32_t* __restrict__ o_first, uint32_t* __restrict__ o_index,
                                                           uint32_t* __restrict__ counters, BvhNode* __restrict__ nodes2, uint32_t* __restrict__ perm, uint32_t n_nodes) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    // (a frontier can only exceed the node count if the links do not form a tree; the host sees the count and fails the build --
    //  nothing is written past the arrays meanwhile)
    bool active = i < counters[kPlocLevelBase + level] && i < n_nodes;
    int32_t l = -1, r = -1;
    uint32_t first = 0, lc = 0;
    if (active) {
        const uint32_t t = f_node[i], me = f_index[i];
        first = f_first[i];
        const uint32_t cnt = t < n_nodes ? t_count[t] : 0u;
        if (t < n_nodes) { l = t_left[t]; r = t_right[t]; }
        lc = l < 0 ? 1u : ((uint32_t)l < n_nodes ? t_count[l] : 0u);
        // links that do not describe a tree over exactly these leaves: flag it, write nothing
        if (t >= n_nodes || me >= n_nodes || cnt < 2u || lc == 0u || lc >= cnt || first + cnt > n_nodes + 1u || (l >= 0 && (uint32_t)l >= n_nodes) ||
            (r >= 0 && (uint32_t)r >= n_nodes) || (l < 0 && (uint32_t)~l > n_nodes) || (r < 0 && (uint32_t)~r > n_nodes)) {
            counters[kPlocLayoutError] = 1u; active = false; l = r = -1;
        }
    }
    if (active) {
        const uint32_t me = f_index[i];
        const uint32_t cnt = t_count[f_node[i]];
        BvhNode& n = nodes2[me];
        n.child0 = l < 0 ? ~(int32_t)first : (int32_t)(first + lc - 1u);
        n.child1 = r < 0 ? ~(int32_t)(first + lc) : (int32_t)(first + lc);
        n._pad[0] = first; n._pad[1] = cnt;
        if (l < 0) perm[first] = (uint32_t)~l;
        if (r < 0) perm[first + lc] = (uint32_t)~r;
    }
    // inner children go to the next level's frontier: one atomic per wave and side
    const uint32_t lane = __lane_id();
    for (int side = 0; side < 2; side++) {
        const bool push = active && (side == 0 ? l >= 0 : r >= 0);
        const unsigned long long m = __ballot(push);
        if (m == 0) continue;
        const int leader = __ffsll((long long)m) - 1;
        uint32_t base = 0;
        if ((int)lane == leader) base = atomicAdd(counters + kPlocLevelBase + level + 1, (uint32_t)__popcll(m));
        base = __shfl(base, leader, 64);
        if (push) {
            const uint32_t at = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (at >= n_nodes) continue;
            o_node[at] = (uint32_t)(side == 0 ? l : r);
            o_first[at] = side == 0 ? first : first + lc;
            o_index[at] = side == 0 ? first + lc - 1u : first + lc;
        }
    }
}

// triangle packets into the final (depth-first leaf) order
__global__ __launch_bounds__(256) void k_ploc_gather(const TriPacket* __restrict__ in, const uint32_t* __restrict__ perm, uint32_t n, TriPacket* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4* s4 = (const float4*)(in + min(perm[i], n - 1u));
    float4* d4 = (float4*)(out + i);
    d4[0] = s4[0]; d4[1] = s4[1]; d4[2] = s4[2];
}

static void free_all(AccelScratch& s) {
    hipFree(s.tris_unsorted); hipFree(s.keys_a); hipFree(s.keys_b); hipFree(s.vals_a); hipFree(s.vals_b);
    hipFree(s.leaf_parent); hipFree(s.node_parent); hipFree(s.seg); hipFree(s.block_bounds); hipFree(s.sort_temp);
    hipFree(s.nodes2); hipFree(s.kept); hipFree(s.widx); hipFree(s.collapse_counters); hipFree(s.wide_ranges);
    hipFree(s.ploc_c[0]); hipFree(s.ploc_c[1]); hipFree(s.ploc_nn); hipFree(s.ploc_valid); hipFree(s.ploc_pos); hipFree(s.ploc_left); hipFree(s.ploc_right);
    hipFree(s.ploc_count); hipFree(s.ploc_perm); hipFree(s.ploc_scan_temp); hipFree(s.ploc_counters);
    hipFree(s.reins_box); hipFree(s.reins_parent); hipFree(s.reins_gain); hipFree(s.reins_out); hipFree(s.reins_top); hipFree(s.reins_lock); hipFree(s.reins_state);
}

static hipError_t ensure(AccelScratch& s, size_t n) {
    if (n <= s.capacity) return hipSuccess;
    free_all(s);
    uint32_t* bounds = s.bounds;
    const int builder = s.builder, passes = s.reinsert_passes;
    const float min_gain = s.reinsert_min_gain;
    const uint32_t fallbacks = s.fallbacks;
    const std::string fallback_why = s.fallback_why;
    s = AccelScratch();
    s.bounds = bounds;
    s.builder = builder; s.reinsert_passes = passes; s.reinsert_min_gain = min_gain;
    s.fallbacks = fallbacks; s.fallback_why = fallback_why;
    size_t cap = n + n / 8 + 64;
    hipError_t e;
    if ((e = hipMalloc(&s.tris_unsorted, cap * sizeof(TriPacket)))) return e;
    if ((e = hipMalloc(&s.keys_a, cap * 8))) return e;
    if ((e = hipMalloc(&s.keys_b, cap * 8))) return e;
    if ((e = hipMalloc(&s.vals_a, cap * 4))) return e;
    if ((e = hipMalloc(&s.vals_b, cap * 4))) return e;
    if ((e = hipMalloc(&s.leaf_parent, cap * 4))) return e;
    if ((e = hipMalloc(&s.node_parent, cap * 4))) return e;
    s.seg_leaves = 256;
    while (s.seg_leaves < cap) s.seg_leaves <<= 1;
    if ((e = hipMalloc(&s.seg, 2 * s.seg_leaves * 32))) return e;
    if ((e = hipMalloc(&s.block_bounds, (cap / 256 + 2) * 6 * sizeof(float)))) return e;
    if ((e = hipMalloc(&s.nodes2, cap * sizeof(BvhNode)))) return e;
    if ((e = hipMalloc(&s.kept, (cap + 1) * 4))) return e;
    if ((e = hipMalloc(&s.widx, (cap + 1) * 4))) return e;
    if (!s.bounds && (e = hipMalloc(&s.bounds, 6 * 4))) return e;
    const size_t tb = radix_sort_temp_bytes(cap);
    if ((e = hipMalloc(&s.sort_temp, tb))) return e;
    s.sort_temp_bytes = tb;
    if ((e = hipMalloc(&s.collapse_counters, (kCollapseMaxLevels + 16) * 4))) return e;
    if ((e = hipMalloc(&s.wide_ranges, cap * sizeof(WideRanges)))) return e;
    // PLOC builder: two cluster arrays, partner / keep / position per cluster, the temporary tree, the final order
    for (int k = 0; k < 2; k++) if ((e = hipMalloc(&s.ploc_c[k], cap * sizeof(PlocCluster)))) return e;
    if ((e = hipMalloc(&s.ploc_nn, cap * 4))) return e;
    if ((e = hipMalloc(&s.ploc_valid, cap * 4))) return e;
    if ((e = hipMalloc(&s.ploc_pos, cap * 4))) return e;
    if ((e = hipMalloc(&s.ploc_left, cap * 4))) return e;
    if ((e = hipMalloc(&s.ploc_right, cap * 4))) return e;
    if ((e = hipMalloc(&s.ploc_count, cap * 4))) return e;
    if ((e = hipMalloc(&s.ploc_perm, cap * 4))) return e;
    if ((e = hipMalloc(&s.ploc_counters, (kCollapseMaxLevels + 16) * 4))) return e;
    // reinsertion (4c): per node slot (2 n - 1 of them) box, parent, best move, path lock
    if ((e = hipMalloc(&s.reins_box, 2 * cap * 32))) return e;
    if ((e = hipMalloc(&s.reins_parent, 2 * cap * 4))) return e;
    if ((e = hipMalloc(&s.reins_gain, 2 * cap * 4))) return e;
    if ((e = hipMalloc(&s.reins_out, 2 * cap * 4))) return e;
    if ((e = hipMalloc(&s.reins_top, 2 * cap * 4))) return e;
    if ((e = hipMalloc(&s.reins_lock, 2 * cap * 8))) return e;
    if ((e = hipMalloc(&s.reins_state, 4 * cap))) return e;
    const size_t sb = exclusive_scan_temp_bytes(cap);
    if ((e = hipMalloc(&s.ploc_scan_temp, sb))) return e;
    s.ploc_scan_bytes = sb;
    s.capacity = cap;
    return hipSuccess;
}

void accel_scratch_free(AccelScratch& s) {
    free_all(s);
    hipFree(s.bounds);
    const int builder = s.builder, passes = s.reinsert_passes;
    const float min_gain = s.reinsert_min_gain;
    s = AccelScratch();
    s.builder = builder; s.reinsert_passes = passes; s.reinsert_min_gain = min_gain;
}

// The segment tree over the sorted triangles' boxes (all levels), from the packets as they stand.
static void seg_build(AccelScratch& s, const TriPacket* d_tris, uint32_t n_tris, hipStream_t stream) {
    SegBox* T = (SegBox*)s.seg;
    const uint32_t P = (uint32_t)s.seg_leaves, g = (n_tris + 255) / 256;
    hipLaunchKernelGGL(k_seg_pass<true>, dim3(g), dim3(256), 0, stream, d_tris, T, P, n_tris);
    uint32_t valid = g;                                          // entries of the level eight above that were written
    for (uint32_t base = P >> 8; base > 1; base >>= 8) {         // (a pass that starts at level `base` ends at the root or 8 levels up)
        hipLaunchKernelGGL(k_seg_pass<false>, dim3((valid + 255) / 256), dim3(256), 0, stream, (const TriPacket*)nullptr, T, base, valid);
        valid = (valid + 255) / 256;
    }
}

// PLOC topology + layout: d_tris holds the Morton-sorted packets on entry and the packets in the tree's depth-first leaf order on
// exit; s.nodes2 holds the binary tree in the same form k_hierarchy leaves it (children, range; boxes come from k_fit).
static hipError_t ploc_build(AccelScratch& s, uint32_t n_tris, TriPacket* d_tris, hipStream_t stream) {
    hipError_t e;
    PlocCluster* c[2] = {(PlocCluster*)s.ploc_c[0], (PlocCluster*)s.ploc_c[1]};
    uint32_t* cnt = s.ploc_counters;
    hipLaunchKernelGGL(k_ploc_init, dim3((n_tris + 255) / 256), dim3(256), 0, stream, (const TriPacket*)d_tris, n_tris, c[0], cnt, (SegBox*)s.reins_box);
    uint32_t bound = n_tris;
    int round = 0;
    for (;;) {
        if (round > 4096) { s.why = "PLOC: more than 4096 clustering rounds"; return hipErrorNotReady; }   // every round merges at least the globally best pair; guards a hang
        for (int k = 0; k < 4; k++, round++) {                       // four rounds per look at the count
            const int a = round & 1;
            const dim3 grid((bound + 255) / 256);
            hipLaunchKernelGGL(k_ploc_nearest, grid, dim3(256), 0, stream, (const PlocCluster*)c[a], (const uint32_t*)(cnt + 4 + a), s.ploc_nn);
            hipLaunchKernelGGL(k_ploc_merge, grid, dim3(256), 0, stream, c[a], (const uint32_t*)(cnt + 4 + a), (const uint32_t*)s.ploc_nn, s.ploc_valid, s.ploc_left,
                               s.ploc_right, s.ploc_count, cnt, (SegBox*)s.reins_box, n_tris);
            if ((e = exclusive_scan_u32(s.ploc_scan_temp, s.ploc_valid, s.ploc_pos, (size_t)bound, stream))) return e;
            hipLaunchKernelGGL(k_ploc_compact, grid, dim3(256), 0, stream, (const PlocCluster*)c[a], (const uint32_t*)(cnt + 4 + a), (const uint32_t*)s.ploc_valid,
                               (const uint32_t*)s.ploc_pos, c[a ^ 1], cnt + 4 + (a ^ 1));
        }
        uint32_t n_cur = 0;
        if ((e = hipMemcpyAsync(&n_cur, cnt + 4 + (round & 1), 4, hipMemcpyDeviceToHost, stream))) return e;
        if ((e = hipStreamSynchronize(stream))) return e;
        if (n_cur == bound && n_cur > 1) { s.why = "PLOC: no pair merged in four rounds at " + std::to_string(n_cur) + " clusters"; return hipErrorNotReady; }
        if (n_cur == 0 || n_cur > bound) { s.why = "PLOC: cluster count " + std::to_string(n_cur) + " after round " + std::to_string(round) + " (bound " + std::to_string(bound) + ")"; return hipErrorNotReady; }
        if (getenv("MIPT_DEBUG_PLOC")) fprintf(stderr, "PLOC round %d: %u clusters\n", round, n_cur);
        if (n_cur == 1) break;
        bound = n_cur;
    }
    // reinsertion passes over the finished topology (4c)
    if (s.builder == 2 && n_tris >= 4) {
        const uint32_t total = 2 * n_tris - 1;
        const dim3 grid((total + 255) / 256);
        hipLaunchKernelGGL(k_reins_parents, dim3((n_tris + 255) / 256), dim3(256), 0, stream, (const int32_t*)s.ploc_left, (const int32_t*)s.ploc_right, n_tris,
                           (const PlocCluster*)c[round & 1], s.reins_parent);
        if ((e = hipMemsetAsync(cnt + 6, 0, 4, stream))) return e;
        for (int pass = 0; pass < s.reinsert_passes; pass++) {
            hipLaunchKernelGGL(k_reins_find, grid, dim3(256), 0, stream, (const float4*)s.reins_box, (const int32_t*)s.reins_parent, (const int32_t*)s.ploc_left,
                               (const int32_t*)s.ploc_right, n_tris, s.reinsert_min_gain, s.reins_gain, s.reins_out, s.reins_top);
            if ((e = hipMemsetAsync(s.reins_state, 0, (size_t)total * 2, stream))) return e;             // state[total] and owned[total], contiguous
            for (int lr = 0; lr < kReinsLockRounds; lr++) {
                if ((e = hipMemsetAsync(s.reins_lock, 0, (size_t)total * 8, stream))) return e;
                hipLaunchKernelGGL(k_reins_lock<true>, grid, dim3(256), 0, stream, (const float*)s.reins_gain, (const uint32_t*)s.reins_out, (const int32_t*)s.reins_top,
                                   (const int32_t*)s.reins_parent, n_tris, s.reins_lock, s.reins_state + total, s.reins_state);
                hipLaunchKernelGGL(k_reins_lock<false>, grid, dim3(256), 0, stream, (const float*)s.reins_gain, (const uint32_t*)s.reins_out, (const int32_t*)s.reins_top,
                                   (const int32_t*)s.reins_parent, n_tris, s.reins_lock, s.reins_state + total, s.reins_state);
            }
            hipLaunchKernelGGL(k_reins_apply, grid, dim3(256), 0, stream, (const uint8_t*)s.reins_state, (const uint32_t*)s.reins_out, (const int32_t*)s.reins_top,
                               s.reins_parent, s.ploc_left, s.ploc_right, s.ploc_count, (float4*)s.reins_box, n_tris, cnt + 6);
            if (getenv("MIPT_DEBUG_PLOC")) {
                uint32_t moved = 0;
                if ((e = hipMemcpyAsync(&moved, cnt + 6, 4, hipMemcpyDeviceToHost, stream))) return e;
                if ((e = hipStreamSynchronize(stream))) return e;
                fprintf(stderr, "reinsertion pass %d: %u moves so far\n", pass + 1, moved);
            }
        }
    }
    // layout: ranges, final indices, final triangle order
    uint32_t* fn[2] = {s.kept, s.vals_a};
    uint32_t* ff[2] = {s.widx, s.vals_b};
    uint32_t* fi[2] = {(uint32_t*)s.leaf_parent, (uint32_t*)s.node_parent};
    hipLaunchKernelGGL(k_ploc_layout_init, dim3(1), dim3(256), 0, stream, (const PlocCluster*)c[round & 1], fn[0], ff[0], fi[0], cnt);
    uint32_t fb = 1;
    const uint32_t n_nodes = n_tris - 1;
    for (uint32_t level = 0, cur = 0;;) {
        if (level + 8 >= (uint32_t)kCollapseMaxLevels) { s.why = "PLOC: tree deeper than " + std::to_string(kCollapseMaxLevels) + " levels"; return hipErrorNotReady; }
        for (int j = 0; j < 8; j++, level++, cur ^= 1u) {
            hipLaunchKernelGGL(k_ploc_layout_level, dim3((fb + 255) / 256), dim3(256), 0, stream, (const int32_t*)s.ploc_left, (const int32_t*)s.ploc_right,
                               (const uint32_t*)s.ploc_count, (const uint32_t*)fn[cur], (const uint32_t*)ff[cur], (const uint32_t*)fi[cur], level, fn[cur ^ 1u], ff[cur ^ 1u],
                               fi[cur ^ 1u], cnt, s.nodes2, s.ploc_perm, n_nodes);
            fb = fb > n_nodes / 2 ? n_nodes : fb * 2;
        }
        uint32_t next = 0;
        uint32_t bad = 0;
        if ((e = hipMemcpyAsync(&next, cnt + kPlocLevelBase + level, 4, hipMemcpyDeviceToHost, stream))) return e;
        if ((e = hipMemcpyAsync(&bad, cnt + kPlocLayoutError, 4, hipMemcpyDeviceToHost, stream))) return e;
        if ((e = hipStreamSynchronize(stream))) return e;
        if (bad) { s.why = "PLOC: the clustered links do not form a tree over the triangles (layout level " + std::to_string(level) + ")"; return hipErrorNotReady; }
        if (next > n_nodes) { s.why = "PLOC: layout frontier of " + std::to_string(next) + " entries at level " + std::to_string(level); return hipErrorNotReady; }
        if (next == 0) break;
        fb = next;
    }
    hipLaunchKernelGGL(k_ploc_gather, dim3((n_tris + 255) / 256), dim3(256), 0, stream, (const TriPacket*)d_tris, (const uint32_t*)s.ploc_perm, n_tris, s.tris_unsorted);
    if ((e = hipMemcpyAsync(d_tris, s.tris_unsorted, (size_t)n_tris * sizeof(TriPacket), hipMemcpyDeviceToDevice, stream))) return e;
    return hipGetLastError();
}

hipError_t accel_build(AccelScratch& s, const BufferRec* d_buffers, const InstanceRec* d_instances, int n_inst, uint32_t n_tris, Bvh4Node* d_nodes,
                       TriPacket* d_tris, ShadePacket* d_shade, int32_t* root_out, uint32_t* wide_nodes_out, uint32_t* stack_need_out, hipStream_t stream) {
    *root_out = 0;
    *wide_nodes_out = 0;
    *stack_need_out = 0;
    if (n_tris == 0) return hipSuccess;
    hipError_t e = ensure(s, n_tris);
    if (e) return e;
    const uint32_t g = (n_tris + 255) / 256;
    hipLaunchKernelGGL(k_setup, dim3(g), dim3(256), 0, stream, d_buffers, d_instances, n_inst, n_tris, s.tris_unsorted, s.block_bounds);
    hipLaunchKernelGGL(k_bounds, dim3(1), dim3(256), 0, stream, s.block_bounds, g, s.bounds);
    if (n_tris == 1) {
        *root_out = ~0;
        if ((e = hipMemcpyAsync(d_tris, s.tris_unsorted, sizeof(TriPacket), hipMemcpyDeviceToDevice, stream))) return e;
        hipLaunchKernelGGL(k_shade_packets, dim3(1), dim3(256), 0, stream, d_tris, 1u, d_instances, d_shade);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_morton, dim3(g), dim3(256), 0, stream, s.tris_unsorted, n_tris, s.bounds, s.keys_a, s.vals_a);
    if ((e = radix_sort_pairs_u64_u32(s.sort_temp, s.keys_a, s.keys_b, s.vals_a, s.vals_b, (size_t)n_tris, stream))) return e;      // sort_scan.hip: stable, 63 bits
    hipLaunchKernelGGL(k_reorder, dim3(g), dim3(256), 0, stream, s.tris_unsorted, s.vals_b, n_tris, d_tris);
    bool radix = s.builder == 0;
    if (!radix) {
        e = ploc_build(s, n_tris, d_tris, stream);
        // Every way the clustering can give up -- no pair merged in four rounds, the round cap (a monotone chain of clusters merges one
        // pair a round), a layout that does not close, a tree deeper than the level cap -- is reported as hipErrorNotReady with s.why set,
        // and all of them happen BEFORE k_ploc_gather rewrites d_tris: the radix tree over the same, still untouched Morton order (keys_b)
        // takes over rather than the build failing.  The caller sees it in s.fallbacks / s.fallback_why.
        if (e == hipErrorNotReady) { radix = true; s.fallbacks++; s.fallback_why = s.why; s.why.clear(); (void)hipGetLastError(); }
        else if (e) return e;
    }
    if (radix) hipLaunchKernelGGL(k_hierarchy, dim3(g), dim3(256), 0, stream, s.keys_b, (int)n_tris, s.nodes2, s.node_parent, s.leaf_parent);
    hipLaunchKernelGGL(k_shade_packets, dim3(g), dim3(256), 0, stream, d_tris, n_tris, d_instances, d_shade);
    const uint32_t n_nodes = n_tris - 1;
    seg_build(s, d_tris, n_tris, stream);
    hipLaunchKernelGGL(k_fit, dim3((n_nodes + 255) / 256), dim3(256), 0, stream, (const SegBox*)s.seg, (uint32_t)s.seg_leaves, n_nodes, s.nodes2);
    // greedy collapse, level by level: frontier = binary nodes that become wide nodes, with the wide index their parent gave them
    uint32_t* fr[2] = {s.kept, s.vals_a};
    uint32_t* wi[2] = {s.widx, s.vals_b};                            // (the sort's value buffers are free again by now)
    uint32_t* nd[2] = {(uint32_t*)s.leaf_parent, (uint32_t*)s.node_parent};   // (k_hierarchy's parent links are not read by anything)
    hipLaunchKernelGGL(k_collapse_init, dim3(1), dim3(256), 0, stream, fr[0], wi[0], nd[0], s.collapse_counters);
    if (n_nodes <= kSmallCollapseNodes) {
        hipLaunchKernelGGL(k_collapse_small, dim3(1), dim3(1024), 0, stream, s.nodes2, fr[0], wi[0], fr[1], wi[1], s.collapse_counters, d_nodes, s.wide_ranges, nd[0], nd[1]);
        // a full build is the rare event now (dynamic geometry is refitted): wait for the node count and the tree's depth, so
        // that a tree too deep for the traversal stack is refused here instead of dropping pushes during a trace
        uint32_t res[2] = {0, 0};
        if ((e = hipMemcpyAsync(&res[0], s.collapse_counters, 4, hipMemcpyDeviceToHost, stream))) return e;
        if ((e = hipMemcpyAsync(&res[1], s.collapse_counters + kCollapseNeedSlot, 4, hipMemcpyDeviceToHost, stream))) return e;
        if ((e = hipStreamSynchronize(stream))) return e;
        *wide_nodes_out = res[0];
        *stack_need_out = res[1];
        return hipGetLastError();
    }
    // Levels are launched eight at a time with grids sized for the largest frontier each can have (four times the one before,
    // at most every node); the kernels read the true sizes from the device, the host looks once per eight levels.
    uint32_t bound = 1;
    for (uint32_t level = 0, cur = 0;;) {
        if (level >= (uint32_t)kCollapseMaxLevels) { s.why = "collapse: more than " + std::to_string(kCollapseMaxLevels) + " levels"; return hipErrorUnknown; }   // guards a hang
        for (int j = 0; j < 8; j++, level++, cur ^= 1u) {
            hipLaunchKernelGGL(k_collapse_level, dim3((bound + 255) / 256), dim3(256), 0, stream, s.nodes2, fr[cur], wi[cur], level, fr[cur ^ 1u], wi[cur ^ 1u],
                               s.collapse_counters, d_nodes, s.wide_ranges, nd[cur], nd[cur ^ 1u]);
            bound = bound > n_nodes / kBvhWidth ? n_nodes : bound * kBvhWidth;
        }
        uint32_t next = 0;
        if ((e = hipMemcpyAsync(&next, s.collapse_counters + 1 + level, 4, hipMemcpyDeviceToHost, stream))) return e;
        if ((e = hipMemcpyAsync(wide_nodes_out, s.collapse_counters, 4, hipMemcpyDeviceToHost, stream))) return e;
        if ((e = hipStreamSynchronize(stream))) return e;
        if (next > n_nodes) { s.why = "collapse: frontier of " + std::to_string(next) + " entries at level " + std::to_string(level); return hipErrorUnknown; }
        if (next == 0) {
            if ((e = hipMemcpy(stack_need_out, s.collapse_counters + kCollapseNeedSlot, 4, hipMemcpyDeviceToHost))) return e;
            break;
        }
        bound = next;
    }
    return hipGetLastError();
}

// ---- refit (UpdateDynamicBlas, Source/RayTracingAccelerationStructure.cpp:110-158, as driven by Pathtracer::UpdateAllBlas,
// Source/Pathtracer.cpp:168-183): the tree's topology, the triangles' sorted order and every untouched packet stay; the packets of the
// instances whose vertices or transform changed are rewritten in place, the boxes follow.
//
// 1. one lane per sorted triangle: if its instance is marked, rebuild its 48-B intersection packet and its 128-B shading packet
__global__ __launch_bounds__(256) void k_refit_packets(const InstanceRec* __restrict__ instances, const uint8_t* __restrict__ touched, uint32_t n_tris,
                                                       TriPacket* __restrict__ tris, ShadePacket* __restrict__ shade) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tris) return;
    const uint32_t inst = tris[i].inst;
    if (!touched[inst]) return;
    const InstanceRec& in = instances[inst];
    const uint32_t prim = tris[i].prim;
    ShadePacket p;
    memset(&p, 0, sizeof(p));
    vec3 w[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        uint32_t v = prim * 3 + k;
        if (in.p_index) v = in.index_is16 ? (uint32_t)((const uint16_t*)in.p_index)[v] : ((const uint32_t*)in.p_index)[v];
        ShadePacket::V& o = p.v[k];
        const float* q = in.p_position + (size_t)v * 3;
        o.pos[0] = q[0]; o.pos[1] = q[1]; o.pos[2] = q[2];
        if (in.p_tangent_space) o.tangent_space = in.p_tangent_space[v];
        if (in.p_texcoord[0]) { float2 t = in.p_texcoord[0][v]; p.uv0[k][0] = t.x; p.uv0[k][1] = t.y; }
        if (in.p_texcoord[1]) { float2 t = in.p_texcoord[1][v]; p.uv1[k][0] = t.x; p.uv1[k][1] = t.y; }
        if (in.p_color) { uint2 c = in.p_color[v]; p.color[k][0] = c.x; p.color[k][1] = c.y; }
        w[k] = mul_point(in.gpu.transform, v3(q[0], q[1], q[2]));      // the expression k_setup evaluates: same bits as a rebuild
    }
    p.inst = inst;
    TriPacket t;
    t.v0[0] = w[0].x; t.v0[1] = w[0].y; t.v0[2] = w[0].z; t.inst = inst;
    const vec3 e1 = w[1] - w[0], e2 = w[2] - w[0];
    t.e1[0] = e1.x; t.e1[1] = e1.y; t.e1[2] = e1.z; t.prim = prim;
    t.e2[0] = e2.x; t.e2[1] = e2.y; t.e2[2] = e2.z; t.flags = in.mask_flags;
    tris[i] = t;
    const float4* s4 = (const float4*)&p;
    float4* d4 = (float4*)(shade + i);
#pragma unroll
    for (int q = 0; q < 8; q++) d4[q] = s4[q];
}

// 3. one lane per wide node: the children's boxes are range queries of the rebuilt segment tree over the ranges the collapse
//    recorded; the node is requantised on its new grid.  Nodes whose ranges hold no changed triangle come out bit-identical.
__global__ __launch_bounds__(256) void k_refit_wide(const SegBox* __restrict__ T, uint32_t P, const WideRanges* __restrict__ ranges,
                                                    const uint32_t* __restrict__ wide_count, Bvh4Node* __restrict__ nodes) {
    const uint32_t wi = blockIdx.x * blockDim.x + threadIdx.x;
    if (wi >= wide_count[0]) return;
    const WideRanges wr = ranges[wi];
    const uint32_t* words = (const uint32_t*)(nodes + wi);
    int32_t ref[kBvhWidth];
    float lo[kBvhWidth][3], hi[kBvhWidth][3];
    int cnt = 0;
    for (int k = 0; k < kBvhWidth; k++) {
        if (wr.count[k] == 0) break;                         // children are packed from slot 0
        ref[k] = (int32_t)words[4 + k];
        seg_query(T, P, wr.first[k], wr.count[k], lo[k], hi[k]);
        cnt++;
    }
    if (cnt) wide_write(nodes + wi, lo, hi, ref, cnt);
}

hipError_t accel_refit(AccelScratch& s, const InstanceRec* d_instances, const uint8_t* d_touched, uint32_t n_tris, uint32_t wide_nodes,
                       Bvh4Node* d_nodes, TriPacket* d_tris, ShadePacket* d_shade, hipStream_t stream) {
    if (n_tris == 0) return hipSuccess;
    if (n_tris > s.capacity) return hipErrorInvalidValue;           // no build has sized the scratch: the caller must rebuild
    const uint32_t g = (n_tris + 255) / 256;
    hipLaunchKernelGGL(k_refit_packets, dim3(g), dim3(256), 0, stream, d_instances, d_touched, n_tris, d_tris, d_shade);
    if (n_tris == 1) return hipGetLastError();                      // a single triangle has no node (root = ~0)
    seg_build(s, d_tris, n_tris, stream);
    hipLaunchKernelGGL(k_refit_wide, dim3((wide_nodes + 255) / 256), dim3(256), 0, stream, (const SegBox*)s.seg, (uint32_t)s.seg_leaves,
                       (const WideRanges*)s.wide_ranges, (const uint32_t*)s.collapse_counters, d_nodes);
    return hipGetLastError();
}

}  // namespace pt
