// accel.hip -- on-device LBVH build for gfx950 (replaces the D3D12 driver's BLAS/TLAS builds,
// Source/RayTracingAccelerationStructure.cpp:157,213,289 as driven by Pathtracer.cpp:138-257).
//
// The reference keeps one BLAS per primitive and rebuilds a TLAS of <= 1000 instances every frame.
// Here every instance is flattened into ONE world-space triangle soup and a single BVH2 is built over
// it (no per-instance ray transform, no overlapping instance boxes to enter):
//   1. setup     : (instance, primitive) -> world-space 48-B packet, centroid bounds by float atomics
//   2. morton    : 63-bit Morton code of the centroid (21 bits / axis)
//   3. sort      : rocPRIM radix sort of (code, triangle id)                       [library primitive]
//   4. hierarchy : Karras 2012 radix tree, one lane per internal node
//   5. fit       : bottom-up AABB propagation, second arriver continues (agent-scope acq_rel counter)
// Every stage streams its arrays once (HBM-bound; 48 B + 8 B + 64 B per triangle).
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>

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

__global__ void k_init_bounds(uint32_t* b) {
    if (threadIdx.x < 3) b[threadIdx.x] = 0xffffffffu;
    else if (threadIdx.x < 6) b[threadIdx.x] = 0u;
}

__device__ __forceinline__ uint32_t load_index(const BufferRec* buffers, int desc, uint32_t i) {
    if (desc == -1) return i;
    const BufferRec& b = buffers[desc];
    return b.format == PT_FORMAT_R16_UINT ? (uint32_t)((const uint16_t*)b.ptr)[i] : ((const uint32_t*)b.ptr)[i];
}

// 1. one lane per triangle: find its instance (binary search on tri_offset), build the world-space packet.
__global__ __launch_bounds__(256) void k_setup(const BufferRec* __restrict__ buffers, const InstanceRec* __restrict__ instances, int n_inst,
                                               uint32_t n_tris, TriPacket* __restrict__ out, uint32_t* __restrict__ bounds) {
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
    // wave-level min/max of the centroid, one atomic per wave per component
    float mnx = valid ? c.x : INFINITY, mny = valid ? c.y : INFINITY, mnz = valid ? c.z : INFINITY;
    float mxx = valid ? c.x : -INFINITY, mxy = valid ? c.y : -INFINITY, mxz = valid ? c.z : -INFINITY;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mnx = fminf(mnx, __shfl_down(mnx, off, 64)); mny = fminf(mny, __shfl_down(mny, off, 64)); mnz = fminf(mnz, __shfl_down(mnz, off, 64));
        mxx = fmaxf(mxx, __shfl_down(mxx, off, 64)); mxy = fmaxf(mxy, __shfl_down(mxy, off, 64)); mxz = fmaxf(mxz, __shfl_down(mxz, off, 64));
    }
    if ((threadIdx.x & 63) == 0 && mnx <= mxx) {
        atomicMin(bounds + 0, float_to_sortable(mnx)); atomicMin(bounds + 1, float_to_sortable(mny)); atomicMin(bounds + 2, float_to_sortable(mnz));
        atomicMax(bounds + 3, float_to_sortable(mxx)); atomicMax(bounds + 4, float_to_sortable(mxy)); atomicMax(bounds + 5, float_to_sortable(mxz));
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
        if (in.p_texcoord[0]) { float2 t = in.p_texcoord[0][v]; o.uv0[0] = t.x; o.uv0[1] = t.y; }
        if (in.p_texcoord[1]) { float2 t = in.p_texcoord[1][v]; o.uv1[0] = t.x; o.uv1[1] = t.y; }
        if (in.p_color) { uint2 c = in.p_color[v]; o.color[0] = c.x; o.color[1] = c.y; }
    }
    const float4* s = (const float4*)&p;
    float4* d = (float4*)(out + i);
#pragma unroll
    for (int q = 0; q < 8; q++) d[q] = s[q];
}

// 5. bottom-up fit.  parent codes are node*2 + slot.
__global__ __launch_bounds__(256) void k_fit(const TriPacket* __restrict__ tris, uint32_t n, BvhNode* nodes, const int32_t* __restrict__ node_parent,
                                             const int32_t* __restrict__ leaf_parent, uint32_t* flags) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const TriPacket& t = tris[i];
    vec3 a = v3p(t.v0), b = a + v3p(t.e1), c = a + v3p(t.e2);
    vec3 lo = hmin(hmin(a, b), c), hi = hmax(hmax(a, b), c);
    int code = leaf_parent[i];
    while (code >= 0) {
        int p = code >> 1, slot = code & 1;
        float* dst = slot ? nodes[p].lo1 : nodes[p].lo0;     // lo then hi are contiguous (6 floats)
        __hip_atomic_store(dst + 0, lo.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(dst + 1, lo.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(dst + 2, lo.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(dst + 3, hi.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(dst + 4, hi.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(dst + 5, hi.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // release my box, acquire the sibling's: the second arriver at a node continues upwards
        uint32_t old = __hip_atomic_fetch_add(flags + p, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old == 0) return;
        const float* src = slot ? nodes[p].lo0 : nodes[p].lo1;
        vec3 slo = v3(__hip_atomic_load(src + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                      __hip_atomic_load(src + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        vec3 shi = v3(__hip_atomic_load(src + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(src + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                      __hip_atomic_load(src + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        lo = hmin(lo, slo); hi = hmax(hi, shi);
        code = node_parent[p];
    }
}

// 6. collapse to 4-wide.  Binary nodes at even depth are kept; each gathers its (up to 4) grandchildren, whose boxes are
//    already stored in the intermediate (odd-depth) nodes.
__global__ __launch_bounds__(256) void k_mark_kept(const int32_t* __restrict__ node_parent, uint32_t n_nodes, uint32_t* __restrict__ kept) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    uint32_t depth = 0;
    int code = node_parent[i];
    while (code >= 0) { depth++; code = node_parent[code >> 1]; }
    kept[i] = (depth & 1u) ? 0u : 1u;
}

__device__ __forceinline__ void wide_set(Bvh4Node& w, int k, const float* lo, const float* hi, int32_t ref) {
    w.lox[k] = lo[0]; w.loy[k] = lo[1]; w.loz[k] = lo[2]; w.hix[k] = hi[0]; w.hiy[k] = hi[1]; w.hiz[k] = hi[2]; w.child[k] = ref;
}

__global__ __launch_bounds__(256) void k_collapse(const BvhNode* __restrict__ nodes2, uint32_t n_nodes, const uint32_t* __restrict__ kept,
                                                  const uint32_t* __restrict__ widx, Bvh4Node* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes || !kept[i]) return;
    const BvhNode& n = nodes2[i];
    Bvh4Node w;
    int k = 0;
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const int32_t ch = c ? n.child1 : n.child0;
        const float* lo = c ? n.lo1 : n.lo0;
        const float* hi = c ? n.hi1 : n.hi0;
        if (ch < 0) wide_set(w, k++, lo, hi, ch);                      // leaf child stays a leaf
        else {                                                         // odd-depth inner node: adopt its two children
            const BvhNode& m = nodes2[ch];
            wide_set(w, k++, m.lo0, m.hi0, m.child0 < 0 ? m.child0 : (int32_t)widx[m.child0]);
            wide_set(w, k++, m.lo1, m.hi1, m.child1 < 0 ? m.child1 : (int32_t)widx[m.child1]);
        }
    }
    const float pinf[3] = {INFINITY, INFINITY, INFINITY}, ninf[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (; k < 4; k++) wide_set(w, k, pinf, ninf, kEmptyChild);
    w._pad[0] = w._pad[1] = w._pad[2] = w._pad[3] = 0;
    const float4* src = (const float4*)&w;
    float4* dst = (float4*)(out + widx[i]);
#pragma unroll
    for (int q = 0; q < 8; q++) dst[q] = src[q];
}

#ifndef PT_GREEDY_COLLAPSE
#define PT_GREEDY_COLLAPSE 1
#endif
constexpr bool kGreedyCollapse = PT_GREEDY_COLLAPSE != 0;

// 6'. collapse to 4-wide, greedily by surface area (one launch per level of the WIDE tree, top down).  A frontier entry is a
//     binary node that becomes a wide node; it starts with its two children and keeps opening the inner child with the largest
//     box until it holds four children or only leaves.  Compared with "keep every even level" this fills the slots (about 3.0 ->
//     3.6 children per node on the Sponza-class scene), so the tree is shallower and a ray visits fewer nodes.
__global__ __launch_bounds__(256) void k_collapse_level(const BvhNode* __restrict__ nodes2, const uint32_t* __restrict__ frontier_in,
                                                        const uint32_t* __restrict__ widx_in, uint32_t n_in, uint32_t* __restrict__ frontier_out,
                                                        uint32_t* __restrict__ widx_out, uint32_t* __restrict__ counters, Bvh4Node* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_in) return;
    // an inner subtree of at most kLeafMax triangles becomes ONE leaf reference (its triangles are contiguous): the bottom of a
    // binary tree is full of 2- and 3-triangle subtrees, which as wide nodes would spend a whole node step on two boxes
    auto child_ref = [&](int32_t r) -> int32_t {
        if (r < 0) return r;
        const uint32_t first = nodes2[r]._pad[0], count = nodes2[r]._pad[1];
        return count <= (uint32_t)kLeafMax ? ~(int32_t)(first | ((count - 1u) << 28)) : r;
    };
    const BvhNode& n = nodes2[frontier_in[i]];
    int32_t ref[4] = {child_ref(n.child0), child_ref(n.child1), kEmptyChild, kEmptyChild};
    float lo[4][3], hi[4][3];
    for (int a = 0; a < 3; a++) { lo[0][a] = n.lo0[a]; hi[0][a] = n.hi0[a]; lo[1][a] = n.lo1[a]; hi[1][a] = n.hi1[a]; }
    int cnt = 2;
    while (cnt < 4) {
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
        ref[pick] = child_ref(m.child0); ref[cnt] = child_ref(m.child1);
        for (int a = 0; a < 3; a++) { lo[pick][a] = m.lo0[a]; hi[pick][a] = m.hi0[a]; lo[cnt][a] = m.lo1[a]; hi[cnt][a] = m.hi1[a]; }
        cnt++;
    }
    Bvh4Node w;
    for (int k = 0; k < 4; k++) {
        if (k < cnt) {
            int32_t r = ref[k];
            if (r >= 0) {                                             // stays an inner node: becomes a wide node of the next level
                const uint32_t wi = atomicAdd(counters + 1, 1u), pos = atomicAdd(counters + 0, 1u);
                frontier_out[pos] = (uint32_t)r;
                widx_out[pos] = wi;
                r = (int32_t)wi;
            }
            wide_set(w, k, lo[k], hi[k], r);
        } else {
            const float pinf[3] = {INFINITY, INFINITY, INFINITY}, ninf[3] = {-INFINITY, -INFINITY, -INFINITY};
            wide_set(w, k, pinf, ninf, kEmptyChild);
        }
    }
    w._pad[0] = w._pad[1] = w._pad[2] = w._pad[3] = 0;
    const float4* src = (const float4*)&w;
    float4* dst = (float4*)(out + widx_in[i]);
#pragma unroll
    for (int q = 0; q < 8; q++) dst[q] = src[q];
}

static void free_all(AccelScratch& s) {
    hipFree(s.tris_unsorted); hipFree(s.keys_a); hipFree(s.keys_b); hipFree(s.vals_a); hipFree(s.vals_b);
    hipFree(s.leaf_parent); hipFree(s.node_parent); hipFree(s.flags); hipFree(s.sort_temp);
    hipFree(s.nodes2); hipFree(s.kept); hipFree(s.widx); hipFree(s.scan_temp); hipFree(s.collapse_counters);
}

static hipError_t ensure(AccelScratch& s, size_t n) {
    if (n <= s.capacity) return hipSuccess;
    free_all(s);
    uint32_t* bounds = s.bounds;
    s = AccelScratch();
    s.bounds = bounds;
    size_t cap = n + n / 8 + 64;
    hipError_t e;
    if ((e = hipMalloc(&s.tris_unsorted, cap * sizeof(TriPacket)))) return e;
    if ((e = hipMalloc(&s.keys_a, cap * 8))) return e;
    if ((e = hipMalloc(&s.keys_b, cap * 8))) return e;
    if ((e = hipMalloc(&s.vals_a, cap * 4))) return e;
    if ((e = hipMalloc(&s.vals_b, cap * 4))) return e;
    if ((e = hipMalloc(&s.leaf_parent, cap * 4))) return e;
    if ((e = hipMalloc(&s.node_parent, cap * 4))) return e;
    if ((e = hipMalloc(&s.flags, cap * 4))) return e;
    if ((e = hipMalloc(&s.nodes2, cap * sizeof(BvhNode)))) return e;
    if ((e = hipMalloc(&s.kept, (cap + 1) * 4))) return e;
    if ((e = hipMalloc(&s.widx, (cap + 1) * 4))) return e;
    if (!s.bounds && (e = hipMalloc(&s.bounds, 6 * 4))) return e;
    size_t tb = 0;
    if ((e = rocprim::radix_sort_pairs(nullptr, tb, s.keys_a, s.keys_b, s.vals_a, s.vals_b, cap, 0, 63, (hipStream_t)0))) return e;
    if ((e = hipMalloc(&s.sort_temp, tb))) return e;
    s.sort_temp_bytes = tb;
    size_t sb = 0;
    if ((e = rocprim::exclusive_scan(nullptr, sb, s.kept, s.widx, 0u, cap + 1, rocprim::plus<uint32_t>(), (hipStream_t)0))) return e;
    if ((e = hipMalloc(&s.scan_temp, sb))) return e;
    s.scan_temp_bytes = sb;
    if ((e = hipMalloc(&s.collapse_counters, 16))) return e;
    s.capacity = cap;
    return hipSuccess;
}

void accel_scratch_free(AccelScratch& s) {
    free_all(s);
    hipFree(s.bounds);
    s = AccelScratch();
}

hipError_t accel_build(AccelScratch& s, const BufferRec* d_buffers, const InstanceRec* d_instances, int n_inst, uint32_t n_tris, Bvh4Node* d_nodes,
                       TriPacket* d_tris, ShadePacket* d_shade, int32_t* root_out, uint32_t* wide_nodes_out, hipStream_t stream) {
    *root_out = 0;
    *wide_nodes_out = 0;
    if (n_tris == 0) return hipSuccess;
    hipError_t e = ensure(s, n_tris);
    if (e) return e;
    const uint32_t g = (n_tris + 255) / 256;
    hipLaunchKernelGGL(k_init_bounds, dim3(1), dim3(64), 0, stream, s.bounds);
    hipLaunchKernelGGL(k_setup, dim3(g), dim3(256), 0, stream, d_buffers, d_instances, n_inst, n_tris, s.tris_unsorted, s.bounds);
    if (n_tris == 1) {
        *root_out = ~0;
        if ((e = hipMemcpyAsync(d_tris, s.tris_unsorted, sizeof(TriPacket), hipMemcpyDeviceToDevice, stream))) return e;
        hipLaunchKernelGGL(k_shade_packets, dim3(1), dim3(256), 0, stream, d_tris, 1u, d_instances, d_shade);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_morton, dim3(g), dim3(256), 0, stream, s.tris_unsorted, n_tris, s.bounds, s.keys_a, s.vals_a);
    size_t tb = s.sort_temp_bytes;
    if ((e = rocprim::radix_sort_pairs(s.sort_temp, tb, s.keys_a, s.keys_b, s.vals_a, s.vals_b, (size_t)n_tris, 0, 63, stream))) return e;
    hipLaunchKernelGGL(k_reorder, dim3(g), dim3(256), 0, stream, s.tris_unsorted, s.vals_b, n_tris, d_tris);
    hipLaunchKernelGGL(k_shade_packets, dim3(g), dim3(256), 0, stream, d_tris, n_tris, d_instances, d_shade);
    if ((e = hipMemsetAsync(s.flags, 0, (size_t)n_tris * 4, stream))) return e;
    hipLaunchKernelGGL(k_hierarchy, dim3(g), dim3(256), 0, stream, s.keys_b, (int)n_tris, s.nodes2, s.node_parent, s.leaf_parent);
    hipLaunchKernelGGL(k_fit, dim3(g), dim3(256), 0, stream, d_tris, n_tris, s.nodes2, s.node_parent, s.leaf_parent, s.flags);
    const uint32_t n_nodes = n_tris - 1;
    if (!kGreedyCollapse) {
        hipLaunchKernelGGL(k_mark_kept, dim3(g), dim3(256), 0, stream, s.node_parent, n_nodes, s.kept);
        if ((e = hipMemsetAsync(s.kept + n_nodes, 0, 4, stream))) return e;            // sentinel: widx[n_nodes] = total kept
        size_t sb = s.scan_temp_bytes;
        if ((e = rocprim::exclusive_scan(s.scan_temp, sb, s.kept, s.widx, 0u, (size_t)n_nodes + 1, rocprim::plus<uint32_t>(), stream))) return e;
        hipLaunchKernelGGL(k_collapse, dim3(g), dim3(256), 0, stream, s.nodes2, n_nodes, s.kept, s.widx, d_nodes);
        if ((e = hipGetLastError())) return e;
        if ((e = hipMemcpyAsync(wide_nodes_out, s.widx + n_nodes, 4, hipMemcpyDeviceToHost, stream))) return e;
        return hipStreamSynchronize(stream);
    }
    // greedy collapse, level by level: frontier = binary nodes that become wide nodes, with the wide index their parent gave them
    uint32_t* fr[2] = {s.kept, s.vals_a};
    uint32_t* wi[2] = {s.widx, s.vals_b};                            // (the sort's value buffers are free again by now)
    const uint32_t first[2] = {0u, 0u};                              // root: binary node 0 -> wide node 0
    if ((e = hipMemcpyAsync(fr[0], &first[0], 4, hipMemcpyHostToDevice, stream))) return e;
    if ((e = hipMemcpyAsync(wi[0], &first[1], 4, hipMemcpyHostToDevice, stream))) return e;
    const uint32_t init[2] = {0u, 1u};                               // [0] next frontier size, [1] wide nodes allocated so far
    if ((e = hipMemcpyAsync(s.collapse_counters, init, 8, hipMemcpyHostToDevice, stream))) return e;
    uint32_t count = 1;
    for (int level = 0, cur = 0; count > 0; level++, cur ^= 1) {
        if (level > 4096) return hipErrorUnknown;                    // a tree cannot be deeper than its node count allows; guards a hang
        hipLaunchKernelGGL(k_collapse_level, dim3((count + 255) / 256), dim3(256), 0, stream, s.nodes2, fr[cur], wi[cur], count, fr[cur ^ 1], wi[cur ^ 1],
                           s.collapse_counters, d_nodes);
        uint32_t c[2];
        if ((e = hipMemcpyAsync(c, s.collapse_counters, 8, hipMemcpyDeviceToHost, stream))) return e;
        if ((e = hipStreamSynchronize(stream))) return e;
        count = c[0];
        *wide_nodes_out = c[1];
        if (count > n_nodes) return hipErrorUnknown;
        if ((e = hipMemsetAsync(s.collapse_counters, 0, 4, stream))) return e;
    }
    return hipGetLastError();
}

}  // namespace pt
