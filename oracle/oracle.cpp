// oracle/oracle.cpp -- TEST INFRASTRUCTURE ONLY.
//
// CPU oracle for the path-tracing hot path of l-johnson-code/glTF-Renderer: a plain C++ restatement
// of Source/Shaders/PathTracer.lib.hlsl (recursive, as the DXR shaders are), Material.hlsli,
// Skin.cs.hlsl, the four environment-map compute shaders, ToneMapper.ps.hlsl and the host loop
// Pathtracer::PathtraceScene (Source/Pathtracer.cpp:259-367).  The DXR pieces that are not in the
// reference's sources (BVH build, traversal, ray/triangle test, texture filtering) are restated
// from the D3D12/DXR functional rules listed in SURVEY.md section 10.
//
// PARITY UNPINNED: the reference cannot be built or run here (Windows/D3D12/DXR, empty submodules)
// and ships no tests, golden images or known-answer vectors for this path.  The only
// reference-supplied fixture is Resources/Sheen_E.exr (tests/golden/sheen_e_16x16.npy).  This oracle
// is therefore pinned by analytic known-answer tests (tests/test_oracle_*.py) and nothing else.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "shading.h"

namespace orc {

// ------------------------------------------------------------------------------------------------
// Host<->device contract types, restated.  Byte-identical to include/mipt.h.
struct TextureAddress {                 // Material.hlsli:14-21
    int descriptor, sampler_index, tex_coord; float rotation; float2 offset, scale;
};
struct Material {                       // Material.hlsli:23-66 (640 B)
    int flags, alpha_mode; float metalness_factor, roughness_factor; float4 base_color_factor;
    float occlusion_factor; float3 emissive_factor; float alpha_cutoff, ior, normal_scale, pad_0;
    TextureAddress normal, albedo, metallic_roughness, occlusion, emissive;
    float specular_factor; float3 specular_color_factor; TextureAddress specular, specular_color;
    float clearcoat_factor, clearcoat_roughness_factor, clearcoat_normal_scale, pad_1;
    TextureAddress clearcoat, clearcoat_roughness, clearcoat_normal;
    float anisotropy_strength, anisotropy_rotation, pad_2[2]; TextureAddress anisotropy;
    float3 sheen_color_factor; float sheen_roughness_factor; TextureAddress sheen_color, sheen_roughness;
    float transmission_factor, thickness_factor, pad_3[2]; TextureAddress transmission;
    float attenuation_distance; float3 attenuation_color; TextureAddress thickness;
};
static_assert(sizeof(TextureAddress) == 32 && sizeof(Material) == 640, "Material layout");
enum { ALPHA_MODE_OPAQUE, ALPHA_MODE_MASK, ALPHA_MODE_BLEND };      // Material.hlsli:8-12
struct Instance {                       // PathTracer.lib.hlsl:32-41 (156 B)
    float4x4 transform, normal_transform;
    int index_descriptor, position_descriptor, tangent_space_descriptor, texcoord_descriptors[2], color_descriptor, material_id;
};
static_assert(sizeof(Instance) == 156, "Instance layout");
struct InstanceDesc {                   // one TLAS instance (Pathtracer.cpp:185-257)
    Instance gpu; uint32_t instance_mask, instance_flags, num_of_vertices, num_of_indices; int dynamic;
};
static_assert(sizeof(InstanceDesc) == 176, "InstanceDesc layout");
enum { INSTANCE_FLAG_TRIANGLE_CULL_DISABLE = 0x1, INSTANCE_FLAG_FORCE_NON_OPAQUE = 0x8 };
struct Settings {                       // Pathtracer.h:70-85 (64 B)
    int min_bounces, max_bounces; uint8_t reset, p0[3]; int debug_output; uint32_t flags;
    float environment_color[3]; float environment_intensity; uint8_t use_frame_as_seed, p1[3]; uint32_t seed;
    float luminance_clamp, min_rr, max_rr; int max_accumulated_frames; float max_ray_length;
};
static_assert(sizeof(Settings) == 64, "Settings layout");
struct ExecuteParams {                  // Pathtracer.h:87-100 as mirrored by include/mipt.h
    float world_to_view[16], view_to_clip[16]; uint32_t width, height; uint64_t frame;
    int light_count, environment_map; void* output; uint32_t tile_rank, tile_rank_count;
};
enum { FMT_R16_UINT = 1, FMT_R32_UINT, FMT_R32G32B32_FLOAT, FMT_R10G10B10A2_UNORM, FMT_R32G32_FLOAT, FMT_R16G16B16A16_UNORM, FMT_JOINT_WEIGHT };
enum DebugOutput {                      // PathTracer.lib.hlsl:43-72
    DBG_NONE, DBG_HIT_KIND, DBG_VERTEX_COLOR, DBG_VERTEX_ALPHA, DBG_VERTEX_NORMAL, DBG_VERTEX_TANGENT, DBG_VERTEX_BITANGENT,
    DBG_TEXCOORD_0, DBG_TEXCOORD_1, DBG_COLOR, DBG_ALPHA, DBG_SHADING_NORMAL, DBG_SHADING_TANGENT, DBG_SHADING_BITANGENT,
    DBG_METALNESS, DBG_ROUGHNESS, DBG_SPECULAR, DBG_SPECULAR_COLOR, DBG_CLEARCOAT, DBG_CLEARCOAT_ROUGHNESS, DBG_CLEARCOAT_NORMAL,
    DBG_TRANSMISSIVE, DBG_BOUNCE_DIRECTION, DBG_BOUNCE_BSDF, DBG_BOUNCE_PDF, DBG_BOUNCE_WEIGHT, DBG_BOUNCE_IS_TRANSMISSION,
    DBG_HEMISPHERE_VIEW_SIDE
};

// ------------------------------------------------------------------------------------------------
// Resources (the "descriptor heap")
struct Buffer { std::vector<uint8_t> data; int format; };
struct Texture { int w, h; bool srgb; std::vector<uint8_t> px; };
struct Sampler { int au, av, minf, magf; };
struct EnvMap {
    int N = 0;                                   // cube face size (EnvironmentMap.cpp:92)
    std::vector<std::vector<uint16_t>> cube;     // per mip: 6*n*n*4 halfs (RGBA16F)
    std::vector<int> cube_n;
    int imp_res = 1024;
    std::vector<std::vector<float>> imp;         // sum pyramid, level 0 = 1024^2
};

// D3D texel addressing for one axis
static inline int address(int i, int n, int mode) {
    if (mode == 0) { int m = i % n; return m < 0 ? m + n : m; }
    if (mode == 1) { int p = 2 * n; int m = i % p; if (m < 0) m += p; return m < n ? m : p - 1 - m; }
    return i < 0 ? 0 : (i >= n ? n - 1 : i);
}
static float g_srgb_lut[256];
static bool g_srgb_init = false;
static void init_srgb() {
    if (g_srgb_init) return;
    for (int i = 0; i < 256; i++) {
        double c = i / 255.0;
        g_srgb_lut[i] = (float)(c <= 0.04045 ? c / 12.92 : pow((c + 0.055) / 1.055, 2.4));
    }
    g_srgb_init = true;
}
static inline float4 texel(const Texture& t, int i, int j) {
    const uint8_t* p = &t.px[((size_t)j * t.w + i) * 4];
    if (t.srgb) return {g_srgb_lut[p[0]], g_srgb_lut[p[1]], g_srgb_lut[p[2]], (float)p[3] / 255.0f};
    return {(float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f, (float)p[3] / 255.0f};
}
static inline float safe_coord(float x) {
    if (!(x == x) || std::isinf(x)) return 0.f;
    return clamp(x, -1.0e9f, 1.0e9f);
}
// Texture2D.SampleLevel(sampler, uv, 0): texel centres at +0.5, sRGB decoded before filtering.
static float4 SampleLevel0(const Texture& t, const Sampler& s, float2 uv) {
    float x = safe_coord(uv.x * (float)t.w), y = safe_coord(uv.y * (float)t.h);
    if (s.magf == 0) {
        int i = address((int)floorf(x), t.w, s.au), j = address((int)floorf(y), t.h, s.av);
        return texel(t, i, j);
    }
    x -= 0.5f; y -= 0.5f;
    float fx0 = floorf(x), fy0 = floorf(y);
    float fx = x - fx0, fy = y - fy0;
    int i0 = address((int)fx0, t.w, s.au), i1 = address((int)fx0 + 1, t.w, s.au);
    int j0 = address((int)fy0, t.h, s.av), j1 = address((int)fy0 + 1, t.h, s.av);
    float w00 = (1 - fx) * (1 - fy), w10 = fx * (1 - fy), w01 = (1 - fx) * fy, w11 = fx * fy;
    return texel(t, i0, j0) * w00 + texel(t, i1, j0) * w10 + texel(t, i0, j1) * w01 + texel(t, i1, j1) * w11;
}

// TextureCube face selection: D3D major-axis table == inverse of CubemapToDirection (Transforms.hlsli:10-50)
static inline void dir_to_face(float3 d, int& face, float& u, float& v) {
    float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z), sc, tc, ma;
    if (ax >= ay && ax >= az) { ma = ax; if (d.x >= 0) { face = 0; sc = -d.z; tc = -d.y; } else { face = 1; sc = d.z; tc = -d.y; } }
    else if (ay >= az) { ma = ay; if (d.y >= 0) { face = 2; sc = d.x; tc = d.z; } else { face = 3; sc = d.x; tc = -d.z; } }
    else { ma = az; if (d.z >= 0) { face = 4; sc = d.x; tc = -d.y; } else { face = 5; sc = -d.x; tc = -d.y; } }
    u = 0.5f * (sc / ma + 1.0f);
    v = 0.5f * (tc / ma + 1.0f);
}
static inline float3 cube_texel(const std::vector<uint16_t>& mip, int n, int face, int i, int j) {
    const uint16_t* p = &mip[(((size_t)face * n + j) * n + i) * 4];
    return {half_to_float(p[0]), half_to_float(p[1]), half_to_float(p[2])};
}
// One bilinear tap with seamless edges: a tap outside the face is re-projected through its
// direction onto the neighbouring face and fetched point-wise there.
static float3 cube_tap(const std::vector<uint16_t>& mip, int n, int face, int i, int j) {
    if (i >= 0 && i < n && j >= 0 && j < n) return cube_texel(mip, n, face, i, j);
    float2 uv = {((float)i + 0.5f) / (float)n, ((float)j + 0.5f) / (float)n};
    float3 d = CubemapToDirection(face, uv);
    int f2; float u, v;
    dir_to_face(d, f2, u, v);
    int ii = (int)floorf(u * (float)n), jj = (int)floorf(v * (float)n);
    ii = ii < 0 ? 0 : (ii >= n ? n - 1 : ii); jj = jj < 0 ? 0 : (jj >= n ? n - 1 : jj);
    return cube_texel(mip, n, f2, ii, jj);
}
static float3 SampleCubeMip(const std::vector<uint16_t>& mip, int n, float3 d) {
    int face; float u, v;
    dir_to_face(d, face, u, v);
    if (!(u == u) || !(v == v)) return {0, 0, 0};
    float x = u * (float)n - 0.5f, y = v * (float)n - 0.5f;
    float fx0 = floorf(x), fy0 = floorf(y);
    float fx = x - fx0, fy = y - fy0;
    int i0 = (int)fx0, j0 = (int)fy0;
    float w00 = (1 - fx) * (1 - fy), w10 = fx * (1 - fy), w01 = (1 - fx) * fy, w11 = fx * fy;
    return cube_tap(mip, n, face, i0, j0) * w00 + cube_tap(mip, n, face, i0 + 1, j0) * w10 +
           cube_tap(mip, n, face, i0, j0 + 1) * w01 + cube_tap(mip, n, face, i0 + 1, j0 + 1) * w11;
}
static float3 SampleCubeLevel(const EnvMap& e, float3 d, float level) {
    int nm = (int)e.cube.size();
    level = clamp(level, 0.f, (float)(nm - 1));
    int l0 = (int)floorf(level), l1 = l0 + 1 < nm ? l0 + 1 : l0;
    float f = level - (float)l0;
    float3 a = SampleCubeMip(e.cube[l0], e.cube_n[l0], d);
    if (f == 0 || l1 == l0) return a;
    float3 b = SampleCubeMip(e.cube[l1], e.cube_n[l1], d);
    return a * (1 - f) + b * f;
}

// ------------------------------------------------------------------------------------------------
// Acceleration structure: CPU LBVH over world-space triangles (Morton -> sort -> radix-tree split
// -> bottom-up fit).  Replaces the driver BVH (RayTracingAccelerationStructure.cpp:157,213,289).
struct Tri { float3 v0, e1, e2; uint32_t inst, prim; uint32_t flags; };   // flags bit0: instance mirrored
struct BNode { float3 lo[2], hi[2]; int child[2]; };                       // child<0: leaf ~tri
struct Bvh {
    std::vector<Tri> tris;
    std::vector<BNode> nodes;
    int root = 0;      // if tris.size()==1 there are no nodes and root = ~0
};
static inline uint64_t expand21(uint64_t v) {
    v &= 0x1fffff;
    v = (v | v << 32) & 0x1f00000000ffffull;
    v = (v | v << 16) & 0x1f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}
struct BuildCtx {
    const std::vector<uint64_t>* keys;
    std::vector<BNode>* nodes;
    const std::vector<Tri>* tris;
};
static inline int common_prefix(const std::vector<uint64_t>& k, int i, int j) {
    uint64_t x = k[i] ^ k[j];
    if (x) return __builtin_clzll(x);
    uint32_t d = (uint32_t)(i ^ j);
    return 64 + (d ? __builtin_clz(d) : 32);
}
static void tri_bounds(const Tri& t, float3& lo, float3& hi) {
    float3 a = t.v0, b = t.v0 + t.e1, c = t.v0 + t.e2;
    lo = hmin(hmin(a, b), c); hi = hmax(hmax(a, b), c);
}
// returns child reference; fills bounds.  `idx` = the node's index: the tree is numbered in pre-order (a range of k triangles has k - 1
// inner nodes, so the left child of node idx is idx + 1 and the right child idx + 1 + (split - first)), which is the order the recursion
// visits them in -- and what lets subtrees be built by different threads into one pre-sized array with the SAME numbering.
static int build_range(BuildCtx& c, int first, int last, int idx, float3& lo, float3& hi) {
    if (first == last) { tri_bounds((*c.tris)[first], lo, hi); return ~first; }
    const auto& k = *c.keys;
    int cp = common_prefix(k, first, last);
    int split = first, step = last - first;
    do {
        step = (step + 1) >> 1;
        int ns = split + step;
        if (ns < last && common_prefix(k, first, ns) > cp) split = ns;
    } while (step > 1);
    float3 l0, h0, l1, h1;
    int c0 = build_range(c, first, split, idx + 1, l0, h0);
    int c1 = build_range(c, split + 1, last, idx + 1 + (split - first), l1, h1);
    BNode& n = (*c.nodes)[idx];
    n.lo[0] = l0; n.hi[0] = h0; n.lo[1] = l1; n.hi[1] = h1; n.child[0] = c0; n.child[1] = c1;
    lo = hmin(l0, l1); hi = hmax(h0, h1);
    return idx;
}
// The same range, its subtrees of at most `grain` triangles handed to `tasks` instead of being descended into (their boxes are filled in
// by the second pass below once the tasks have run).
struct SubtreeTask { int first, last, idx; float3 lo, hi; };
static int split_of(const std::vector<uint64_t>& k, int first, int last) {
    int cp = common_prefix(k, first, last);
    int split = first, step = last - first;
    do {
        step = (step + 1) >> 1;
        int ns = split + step;
        if (ns < last && common_prefix(k, first, ns) > cp) split = ns;
    } while (step > 1);
    return split;
}
static void collect_tasks(BuildCtx& c, int first, int last, int idx, int grain, std::vector<SubtreeTask>& tasks) {
    if (first == last || last - first + 1 <= grain) { tasks.push_back({first, last, idx, {}, {}}); return; }
    int split = split_of(*c.keys, first, last);
    collect_tasks(c, first, split, idx + 1, grain, tasks);
    collect_tasks(c, split + 1, last, idx + 1 + (split - first), grain, tasks);
}
static int finish_top(BuildCtx& c, int first, int last, int idx, int grain, const std::vector<SubtreeTask>& tasks, size_t& next, float3& lo, float3& hi) {
    if (first == last || last - first + 1 <= grain) {
        const SubtreeTask& t = tasks[next++];
        lo = t.lo; hi = t.hi;
        return first == last ? ~first : idx;
    }
    int split = split_of(*c.keys, first, last);
    float3 l0, h0, l1, h1;
    int c0 = finish_top(c, first, split, idx + 1, grain, tasks, next, l0, h0);
    int c1 = finish_top(c, split + 1, last, idx + 1 + (split - first), grain, tasks, next, l1, h1);
    BNode& n = (*c.nodes)[idx];
    n.lo[0] = l0; n.hi[0] = h0; n.lo[1] = l1; n.hi[1] = h1; n.child[0] = c0; n.child[1] = c1;
    lo = hmin(l0, l1); hi = hmax(h0, h1);
    return idx;
}
template <class F> static void parallel_chunks(size_t n, int nthreads, F f) {        // f(begin, end, chunk index)
    if (nthreads <= 1 || n < 4096) { f((size_t)0, n, 0); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++) th.emplace_back([=]() { f(n * t / nthreads, n * (t + 1) / nthreads, t); });
    for (auto& t : th) t.join();
}
// nthreads <= 1: the single-threaded build every parity test runs.  nthreads > 1: the same tree, node for node (same Morton codes, a
// stable sort by chunks + stable merges, pre-order numbering), built on that many cores -- bench.py's cpu_baseline leg B2.
static void build_lbvh(Bvh& bvh, int nthreads = 1) {
    const bool prof = getenv("ORC_PROFILE") != nullptr;
    auto tp = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) { if (prof) { auto t = std::chrono::steady_clock::now(); fprintf(stderr, "  lbvh %-10s %.2f ms\n", what, std::chrono::duration<double, std::milli>(t - tp).count()); tp = t; } };
    size_t n = bvh.tris.size();
    bvh.nodes.clear();
    if (n == 0) { bvh.root = 0; return; }
    std::vector<float3> cen(n);
    const int T = std::max(1, nthreads);
    std::vector<float3> plo((size_t)T, F3(INFINITY)), phi((size_t)T, F3(-INFINITY));
    parallel_chunks(n, T, [&](size_t b, size_t e, int t) {
        float3 clo = F3(INFINITY), chi = F3(-INFINITY);
        for (size_t i = b; i < e; i++) {
            float3 lo, hi; tri_bounds(bvh.tris[i], lo, hi);
            cen[i] = (lo + hi) * 0.5f;
            clo = hmin(clo, cen[i]); chi = hmax(chi, cen[i]);
        }
        plo[(size_t)t] = clo; phi[(size_t)t] = chi;
    });
    float3 clo = F3(INFINITY), chi = F3(-INFINITY);
    for (int t = 0; t < T; t++) { clo = hmin(clo, plo[(size_t)t]); chi = hmax(chi, phi[(size_t)t]); }
    float3 ext = chi - clo;
    lap("bounds");
    std::vector<std::pair<uint64_t, uint32_t>> keyed(n);
    auto by_key = [](const std::pair<uint64_t, uint32_t>& a, const std::pair<uint64_t, uint32_t>& b) { return a.first < b.first; };
    std::vector<size_t> cut;
    parallel_chunks(n, T, [&](size_t b, size_t e, int) {
        for (size_t i = b; i < e; i++) {
            float3 q = (cen[i] - clo) / hmax(ext, F3(1e-30f));
            uint64_t x = (uint64_t)clamp(q.x * 2097152.f, 0.f, 2097151.f), y = (uint64_t)clamp(q.y * 2097152.f, 0.f, 2097151.f),
                     z = (uint64_t)clamp(q.z * 2097152.f, 0.f, 2097151.f);
            keyed[i] = {expand21(x) << 2 | expand21(y) << 1 | expand21(z), (uint32_t)i};
        }
        std::stable_sort(keyed.begin() + (ptrdiff_t)b, keyed.begin() + (ptrdiff_t)e, by_key);
    });
    lap("sort");
    if (T > 1 && n >= 4096) {                                         // stable pairwise merges of the sorted chunks, a round per doubling
        for (int t = 0; t <= T; t++) cut.push_back(n * (size_t)t / (size_t)T);
        while (cut.size() > 2) {
            std::vector<std::thread> th;
            std::vector<size_t> nc;
            for (size_t k = 0; k + 2 < cut.size() + 1 && k + 1 < cut.size(); k += 2) {
                const size_t a = cut[k], m = cut[k + 1], e = k + 2 < cut.size() ? cut[k + 2] : cut[k + 1];
                nc.push_back(a);
                if (e > m) th.emplace_back([&keyed, a, m, e, by_key]() { std::inplace_merge(keyed.begin() + (ptrdiff_t)a, keyed.begin() + (ptrdiff_t)m, keyed.begin() + (ptrdiff_t)e, by_key); });
            }
            nc.push_back(n);
            for (auto& t : th) t.join();
            cut.swap(nc);
        }
    }
    lap("merge");
    std::vector<Tri> sorted(n);
    std::vector<uint64_t> keys(n);
    parallel_chunks(n, T, [&](size_t b, size_t e, int) { for (size_t i = b; i < e; i++) { sorted[i] = bvh.tris[keyed[i].second]; keys[i] = keyed[i].first; } });
    bvh.tris.swap(sorted);
    bvh.nodes.assign(n > 1 ? n - 1 : 0, BNode());
    BuildCtx c{&keys, &bvh.nodes, &bvh.tris};
    float3 lo, hi;
    lap("gather");
    if (T <= 1 || n < 4096) { bvh.root = build_range(c, 0, (int)n - 1, 0, lo, hi); lap("tree"); return; }
    std::vector<SubtreeTask> tasks;
    const int grain = (int)std::max<size_t>(256, n / ((size_t)T * 8));
    collect_tasks(c, 0, (int)n - 1, 0, grain, tasks);
    std::atomic<size_t> next_task{0};
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) th.emplace_back([&]() {
        for (;;) {
            size_t k = next_task.fetch_add(1);
            if (k >= tasks.size()) break;
            BuildCtx cc = c;
            build_range(cc, tasks[k].first, tasks[k].last, tasks[k].idx, tasks[k].lo, tasks[k].hi);
        }
    });
    for (auto& t : th) t.join();
    size_t next = 0;
    bvh.root = finish_top(c, 0, (int)n - 1, 0, grain, tasks, next, lo, hi);
    lap("tree");
}

// ------------------------------------------------------------------------------------------------
struct Counters { std::atomic<uint64_t> primary{0}, bounce{0}, shadow{0}, nodes{0}, tris{0}, hits{0}, taps{0}; };
// per-thread tallies, merged into Oracle::counters when a worker finishes (no shared cache line in the hot loop)
struct Tally { uint64_t primary = 0, bounce = 0, shadow = 0, nodes = 0, tris = 0, hits = 0, taps = 0; };
static thread_local Tally t_tally;
static void merge_tally(Counters& c) {
    c.primary += t_tally.primary; c.bounce += t_tally.bounce; c.shadow += t_tally.shadow; c.nodes += t_tally.nodes;
    c.tris += t_tally.tris; c.hits += t_tally.hits; c.taps += t_tally.taps;
    t_tally = Tally();
}

struct Oracle {
    SheenLut lut;
    std::vector<Buffer> buffers;
    std::vector<Texture> textures;
    std::vector<Sampler> samplers;
    std::vector<Material> materials;
    std::vector<Light> lights;
    std::vector<InstanceDesc> instances;
    std::vector<EnvMap> envs;
    Bvh bvh;
    bool accel_dirty = true;
    bool brute_force = false;
    // diagnostics (tools/diag_flip.py): trace only the pixels of a window; record every ray of one pixel with what it found
    uint32_t win[4] = {0, 0, 0xffffffffu, 0xffffffffu};             // x0, y0, x1, y1 (exclusive)
    int64_t log_px = -1, log_py = -1;
    std::vector<float> ray_log;                                     // 16 floats per ray: origin, tmin, direction, tmax, mode, hit, t, instance, primitive, transmission, ray flags, 0
    int bounce_limit = 5;                // Pathtracer::MAX_BOUNCES, Pathtracer.h:102
    // cross-frame state of Pathtracer (Pathtracer.h:152-153)
    float previous_world_to_clip[16] = {0};
    int accumulated_frames = 0;
    Counters counters;
    double last_accel_ms = 0, last_trace_ms = 0;
};

// SceneConstants, PathTracer.lib.hlsl:10-30
struct SceneConstants {
    float4x4 clip_to_world; float3 camera_pos; int num_of_lights; uint32_t res_x, res_y, seed; int accumulated_frames;
    float3 environment_color; float environment_intensity; int debug_output; uint32_t flags; float max_ray_length;
    int min_bounces, max_bounces; int env; float luminance_clamp, min_rr, max_rr;
};

// ---- vertex fetch (PathTracer.lib.hlsl:176-257) -------------------------------------------------
static inline void GetIndices(const Oracle& o, int index_descriptor, uint32_t prim, uint32_t v[3]) {   // :176-184
    v[0] = prim * 3; v[1] = prim * 3 + 1; v[2] = prim * 3 + 2;
    if (index_descriptor != -1) {
        const Buffer& b = o.buffers[index_descriptor];
        for (int i = 0; i < 3; i++) {
            if (b.format == FMT_R16_UINT) v[i] = ((const uint16_t*)b.data.data())[v[i]];
            else v[i] = ((const uint32_t*)b.data.data())[v[i]];
        }
    }
}
static inline float3 fetch_pos(const Oracle& o, int desc, uint32_t v) { return ((const float3*)o.buffers[desc].data.data())[v]; }
template <typename T> static inline T Bary(T a0, T a1, T a2, float3 w) { return w.x * a0 + w.y * a1 + w.z * a2; }   // :155-159
static inline float3 GenerateTangent(float3 n) {                                                        // :166-174
    float3 helper = {1, 0, 0};
    if (fabsf(n.x) > fabsf(n.y)) helper = {0, 1, 0};
    return normalize(cross(helper, n));
}
static inline float4 GetVertexColor(const Oracle& o, int desc, const uint32_t v[3], float3 w) {           // :229-242
    if (desc == -1) return F4(1);
    const uint16_t* p = (const uint16_t*)o.buffers[desc].data.data();
    float4 c[3];
    for (int i = 0; i < 3; i++) c[i] = {p[v[i] * 4] / 65535.f, p[v[i] * 4 + 1] / 65535.f, p[v[i] * 4 + 2] / 65535.f, p[v[i] * 4 + 3] / 65535.f};
    return Bary(c[0], c[1], c[2], w);
}
static inline float2 GetTexcoord(const Oracle& o, int desc, const uint32_t v[3], float3 w) {              // :244-257
    if (desc == -1) return {0, 0};
    const float2* p = (const float2*)o.buffers[desc].data.data();
    return Bary(p[v[0]], p[v[1]], p[v[2]], w);
}
struct VertexAttributes { float3 position, geometric_normal, normal; float4 tangent; float3 bitangent; float4 color; float2 texcoords[2]; };
static VertexAttributes GetVertexAttributes(const Oracle& o, const Instance& in, uint32_t prim, float3 w) {   // :280-302
    VertexAttributes a;
    uint32_t v[3];
    GetIndices(o, in.index_descriptor, prim, v);
    float3 p0 = fetch_pos(o, in.position_descriptor, v[0]), p1 = fetch_pos(o, in.position_descriptor, v[1]), p2 = fetch_pos(o, in.position_descriptor, v[2]);
    a.position = Bary(p0, p1, p2, w);
    a.geometric_normal = cross(p1 - p0, p2 - p0);                              // :196-199 (un-normalised)
    if (in.tangent_space_descriptor != -1) {                                   // :201-222
        const uint32_t* ts = (const uint32_t*)o.buffers[in.tangent_space_descriptor].data.data();
        float3 n[3]; float4 t[3];
        for (int i = 0; i < 3; i++) DecodeTangentSpace(UnpackR10G10B10A2(ts[v[i]]), n[i], t[i]);
        a.normal = Bary(n[0], n[1], n[2], w);
        float3 t3 = Bary(xyz(t[0]), xyz(t[1]), xyz(t[2]), w);
        a.tangent = {t3.x, t3.y, t3.z, t[0].w};                                // winding from vertex 0 only (quirk q16)
    } else {
        a.normal = a.geometric_normal;
        float3 t3 = GenerateTangent(a.geometric_normal);
        a.tangent = {t3.x, t3.y, t3.z, 1};
    }
    a.position = xyz(mul(in.transform, F4(a.position, 1)));
    a.geometric_normal = normalize(xyz(mul(in.normal_transform, F4(a.geometric_normal, 0))));
    a.normal = normalize(xyz(mul(in.normal_transform, F4(a.normal, 0))));
    float3 tw = normalize(xyz(mul(in.transform, F4(xyz(a.tangent), 0))));
    a.tangent = {tw.x, tw.y, tw.z, a.tangent.w};
    a.bitangent = a.tangent.w * normalize(cross(a.normal, xyz(a.tangent)));   // :224-227
    a.color = GetVertexColor(o, in.color_descriptor, v, w);
    for (int i = 0; i < 2; i++) a.texcoords[i] = GetTexcoord(o, in.texcoord_descriptors[i], v, w);
    return a;
}

// ---- Material.hlsli ----------------------------------------------------------------------------
static inline float2 TransformUv(const TextureAddress& a, float2 uv) {          // :68-88
    float3x3 T = M3({1, 0, a.offset.x}, {0, 1, a.offset.y}, {0, 0, 1});
    float c = o_cos(a.rotation), s = o_sin(a.rotation);
    float3x3 R = M3({c, s, 0}, {-s, c, 0}, {0, 0, 1});
    float3x3 S = M3({a.scale.x, 0, 0}, {0, a.scale.y, 0}, {0, 0, 1});
    float3x3 M = mul(T, mul(R, S));
    float3 r = mul(M, float3{uv.x, uv.y, 1});
    return {r.x, r.y};
}
static float4 SampleTexture(Oracle& o, const TextureAddress& a, const float2 tc[2]) {   // :90-96
    float2 uv = TransformUv(a, tc[a.tex_coord]);
    t_tally.taps++;
    return SampleLevel0(o.textures[a.descriptor], o.samplers[a.sampler_index], uv);
}
static float4 GetBaseColor(Oracle& o, const Material& m, const float2 tc[2], float4 vc) {   // :98-106
    float4 c = m.base_color_factor;
    c = c * vc;
    if (m.albedo.descriptor != -1) c = c * SampleTexture(o, m.albedo, tc);
    return c;
}
static inline float GetAlpha(const Material& m, float4 c) {                     // :108-117
    if (m.alpha_mode == ALPHA_MODE_BLEND) return c.w;
    if (m.alpha_mode == ALPHA_MODE_MASK) return c.w < m.alpha_cutoff ? 0.f : 1.f;
    return 1;
}
static float3 NormalFromMap(Oracle& o, const TextureAddress& a, float scale, const float2 tc[2], float3 geometric, const float3x3& t2w) {  // :119-128, :199-208
    if (a.descriptor == -1) return geometric;
    float4 s = SampleTexture(o, a, tc);
    float3 nm = float3{s.x, s.y, s.z} * 2.f - 1.f;
    nm.x *= scale; nm.y *= scale;
    return normalize(mul(t2w, nm));
}
static float3 GetEmissive(Oracle& o, const Material& m, const float2 tc[2]) {   // :151-159
    float3 e = m.emissive_factor;
    if (m.emissive.descriptor != -1) e = e * xyz(SampleTexture(o, m.emissive, tc));
    return e;
}
static inline float3x3 TangentToWorldMatrix(float3 n, float3 t, float3 b) { return transpose(M3(t, b, n)); }   // :272-280

// PathTracer.lib.hlsl:318-381
static SurfaceProperties GetSurfaceProperties(Oracle& o, const SceneConstants& sc, const Material& m, const VertexAttributes& a, float3 view) {
    float3x3 t2w = TangentToWorldMatrix(a.normal, xyz(a.tangent), a.bitangent);
    SurfaceProperties sp;
    float4 base = GetBaseColor(o, m, a.texcoords, a.color);
    sp.albedo = xyz(base);
    sp.alpha = GetAlpha(m, base);
    sp.shading_normal = NormalFromMap(o, m.normal, m.normal_scale, a.texcoords, a.normal, t2w);
    if (sc.flags & F_SHADING_NORMAL_ADAPTATION) sp.shading_normal = NormalAdaptation(a.geometric_normal, sp.shading_normal, view);
    float metal = m.metalness_factor, rough = m.roughness_factor;               // Material.hlsli:130-140
    if (m.metallic_roughness.descriptor != -1) { float4 s = SampleTexture(o, m.metallic_roughness, a.texcoords); metal *= s.z; rough *= s.y; }
    sp.metalness = metal;
    sp.roughness_squared.y = hmax(rough * rough, MINIMUM_ROUGHNESS);
    if (m.occlusion.descriptor != -1) (void)SampleTexture(o, m.occlusion, a.texcoords);   // :339 dead value, fetch still issued
    (void)GetEmissive(o, m, a.texcoords);                                                 // :341 dead value
    sp.ior = m.ior;
    sp.specular_factor = m.specular_factor;                                     // Material.hlsli:161-168
    if (m.specular.descriptor != -1) sp.specular_factor *= SampleTexture(o, m.specular, a.texcoords).w;
    sp.specular_color = m.specular_color_factor;                                // :170-177
    if (m.specular_color.descriptor != -1) sp.specular_color = sp.specular_color * xyz(SampleTexture(o, m.specular_color, a.texcoords));
    sp.clearcoat = m.clearcoat_factor;                                          // :179-186
    if (m.clearcoat.descriptor != -1) sp.clearcoat *= SampleTexture(o, m.clearcoat, a.texcoords).x;
    sp.clearcoat_roughness = m.clearcoat_roughness_factor;                      // :188-195
    if (m.clearcoat_roughness.descriptor != -1) sp.clearcoat_roughness *= SampleTexture(o, m.clearcoat_roughness, a.texcoords).y;
    sp.clearcoat_normal = NormalFromMap(o, m.clearcoat_normal, m.clearcoat_normal_scale, a.texcoords, a.normal, t2w);
    if (sc.flags & F_SHADING_NORMAL_ADAPTATION) sp.clearcoat_normal = NormalAdaptation(a.geometric_normal, sp.clearcoat_normal, view);
    // GetAnisotropyStrengthAndDirection, Material.hlsli:246-262
    float strength = m.anisotropy_strength, rot = m.anisotropy_rotation;
    float3 av = {1, 0, 1};
    if (m.anisotropy.descriptor != -1) {
        float4 s = SampleTexture(o, m.anisotropy, a.texcoords);
        av = {s.x * 2 - 1, s.y * 2 - 1, s.z};
    }
    float cr = o_cos(rot), sr = o_sin(rot);
    float2 adir = normalize(float2{cr * av.x + -sr * av.y, sr * av.x + cr * av.y});
    strength *= av.z;
    // CalculateShadingTangentAndBitangent, Material.hlsli:264-270
    float3 sb = normalize(cross(sp.shading_normal, xyz(a.tangent)));
    float3 st = normalize(cross(sb, sp.shading_normal));
    sb *= a.tangent.w;
    float3x3 st2w = TangentToWorldMatrix(sp.shading_normal, st, sb);
    sp.anisotropy_tangent = normalize(mul(st2w, float3{adir.x, adir.y, 0}));
    sp.anisotropy_bitangent = normalize(cross(sp.anisotropy_tangent, sp.shading_normal));
    sp.roughness_squared.x = hmax(lerp(sp.roughness_squared.y, 1, strength * strength), MINIMUM_ROUGHNESS);
    sp.sheen_color = m.sheen_color_factor;                                      // Material.hlsli:210-217
    if (m.sheen_color.descriptor != -1) sp.sheen_color = sp.sheen_color * xyz(SampleTexture(o, m.sheen_color, a.texcoords));
    float sheen_rough = m.sheen_roughness_factor;                               // :219-226
    if (m.sheen_roughness.descriptor != -1) sheen_rough *= SampleTexture(o, m.sheen_roughness, a.texcoords).w;
    sp.sheen_roughness_squared = hmax(sheen_rough * sheen_rough, MINIMUM_ROUGHNESS);
    sp.transmissive = m.transmission_factor;                                    // :228-235
    if (m.transmission.descriptor != -1) sp.transmissive *= SampleTexture(o, m.transmission, a.texcoords).x;
    sp.thickness = m.thickness_factor;                                          // :237-244
    if (m.thickness.descriptor != -1) sp.thickness *= SampleTexture(o, m.thickness, a.texcoords).y;
    sp.attenuation_distance = m.attenuation_distance;
    sp.attenuation_color = m.attenuation_color;
    return sp;
}

// ---- Sampling.hlsli:123-174 on the importance pyramid --------------------------------------------
static inline float imp_load(const EnvMap& e, int level, uint32_t x, uint32_t y) {
    uint32_t n = (uint32_t)e.imp_res >> level;
    if (x >= n || y >= n) return 0.f;             // out-of-range Load returns 0
    return e.imp[level][(size_t)y * n + x];
}
static float2 SampleImportanceMap(const EnvMap& e, float2 u, float& pdf) {     // :123-163
    uint32_t width = e.imp_res, height = e.imp_res; int mips = (int)e.imp.size();
    pdf = 1;
    uint32_t px = 0, py = 0;
    for (int i = mips - 2; i >= 0; i--) {
        px <<= 1; py <<= 1;
        float ul = imp_load(e, i, px, py), ur = imp_load(e, i, px + 1, py), ll = imp_load(e, i, px, py + 1), lr = imp_load(e, i, px + 1, py + 1);
        float left = ul + ll, right = ur + lr, total = left + right;
        float prob_left = left / total;
        if (u.x < prob_left) {
            u.x /= prob_left;
            float prob_upper = ul / left;
            if (u.y < prob_upper) u.y /= prob_upper;
            else { py++; u.y = (u.y - prob_upper) / (1 - prob_upper); }
        } else {
            px++;
            u.x = (u.x - prob_left) / (1 - prob_left);
            float prob_upper = ur / right;
            if (u.y < prob_upper) u.y /= prob_upper;
            else { py++; u.y = (u.y - prob_upper) / (1 - prob_upper); }
        }
    }
    pdf = (float)width * (float)height * imp_load(e, 0, px, py) / imp_load(e, mips - 1, 0, 0);
    return {((float)px + u.x) / (float)width, ((float)py + u.y) / (float)width};   // both / width (quirk q10)
}
static float ImportanceMapPdf(const EnvMap& e, float2 uv) {                    // :165-174
    int mips = (int)e.imp.size();
    float total = imp_load(e, mips - 1, 0, 0);
    int2 p = UVToPixel(uv, {e.imp_res, e.imp_res});
    float value = (p.x < 0 || p.y < 0) ? 0.f : imp_load(e, 0, (uint32_t)p.x, (uint32_t)p.y);
    return (float)e.imp_res * (float)e.imp_res * value / total;
}

// ------------------------------------------------------------------------------------------------
// TraceRay (DXR semantics)
enum { RAY_FLAG_FORCE_NON_OPAQUE = 0x2, RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH = 0x4, RAY_FLAG_SKIP_CLOSEST_HIT_SHADER = 0x8,
       RAY_FLAG_CULL_BACK_FACING_TRIANGLES = 0x10, RAY_FLAG_CULL_FRONT_FACING_TRIANGLES = 0x20 };
struct RayDesc { float3 origin; float tmin; float3 direction; float tmax; };
struct Hit { float t; float u, v; int tri; bool front; };

struct Payload { float3 throughput; float bsdf_pdf; float3 color; uint32_t flags; int bounce; int random_count; };   // :110-117
enum { PAYLOAD_FLAG_MIS = 1 };

struct Tracer {
    Oracle& o; const SceneConstants& sc; uint32_t px, py;
    const EnvMap* env;

    // Ray/triangle: Moeller-Trumbore on world-space (v0, e1, e2); hit interval tmin < t < tmax.
    // Moeller-Trumbore in float, plus the two rules that make the answer independent of the tree that is walked (csrc/pt_traverse.h
    // candidate_stands states them the same way; the exhaustive search `brute_force` and the LBVH below then find the same hits, which
    // tests/test_gpu_round3.py checks ray by ray):
    //  * the box gate: the float test accepts rays that pass a few ulp outside the triangle, sometimes outside its box, and a tree may or may
    //    not have culled such a ray before the triangle is asked.  A candidate therefore has to pass the node test's box arithmetic on ITS OWN
    //    box (tri_bounds), with a distance consistent with it; every ancestor box contains that box and the test is monotone in the planes,
    //    so no ancestor culls a candidate that stands;
    //  * tie_with: the triangle that holds the current closest hit at distance tmax (or null): a candidate at EXACTLY that distance replaces
    //    it when its (instance, primitive) is lower -- DXR leaves equal distances to the traversal order.
    inline bool intersect(const Tri& t, const RayDesc& r, const float3& inv, float tmax, float& ot, float& ou, float& ov, bool& front, const Tri* tie_with = nullptr) const {
        float3 p = cross(r.direction, t.e2);
        float det = dot(t.e1, p);
        if (det == 0.0f || !(det == det)) return false;
        float invd = 1.0f / det;
        float3 tv = r.origin - t.v0;
        float u = dot(tv, p) * invd;
        if (!(u >= 0.0f) || u > 1.0f) return false;
        float3 q = cross(tv, t.e1);
        float v = dot(r.direction, q) * invd;
        if (!(v >= 0.0f) || u + v > 1.0f) return false;
        float tt = dot(t.e2, q) * invd;
        if (!(tt > r.tmin) || !(tt <= tmax)) return false;
        float3 lo, hi; tri_bounds(t, lo, hi);
        float3 t0 = (lo - r.origin) * inv, t1 = (hi - r.origin) * inv;
        float tn = hmax(hmax(hmin(t0.x, t1.x), hmin(t0.y, t1.y)), hmax(hmin(t0.z, t1.z), r.tmin));
        float tx = hmin(hmin(hmax(t0.x, t1.x), hmax(t0.y, t1.y)), hmax(t0.z, t1.z));
        if (!(tn <= tx * 1.0000004f && tn <= tt * 1.0000004f)) return false;
        if (!(tt < tmax)) {
            if (!(tie_with && (t.inst < tie_with->inst || (t.inst == tie_with->inst && t.prim < tie_with->prim)))) return false;
        }
        ot = tt; ou = u; ov = v;
        front = (det > 0.0f) != ((t.flags & 1) != 0);      // object-space winding (mirrored instances flip)
        return true;
    }
    // AnyHit, PathTracer.lib.hlsl:1010-1035: true = accept.
    bool any_hit_alpha_test(const Tri& t, float u, float v) {
        const Instance& in = o.instances[t.inst].gpu;
        const Material& m = o.materials[in.material_id];
        float3 w = {1 - u - v, u, v};
        uint32_t vi[3];
        GetIndices(o, in.index_descriptor, t.prim, vi);
        float4 base = GetVertexColor(o, in.color_descriptor, vi, w);
        float2 tc[2];
        for (int i = 0; i < 2; i++) tc[i] = GetTexcoord(o, in.texcoord_descriptors[i], vi, w);
        base = GetBaseColor(o, m, tc, base);
        return !(base.w < m.alpha_cutoff);
    }
    // ShadowAnyHit, :1053-1079: returns alpha
    float shadow_alpha(const Tri& t, float u, float v) {
        const Instance& in = o.instances[t.inst].gpu;
        const Material& m = o.materials[in.material_id];
        float3 w = {1 - u - v, u, v};
        uint32_t vi[3];
        GetIndices(o, in.index_descriptor, t.prim, vi);
        float4 base = GetVertexColor(o, in.color_descriptor, vi, w);
        float2 tc[2];
        for (int i = 0; i < 2; i++) tc[i] = GetTexcoord(o, in.texcoord_descriptors[i], vi, w);
        base = GetBaseColor(o, m, tc, base);
        return GetAlpha(m, base);
    }
    // mode 0: closest hit (hit group 0).  mode 1: shadow / occlusion (hit group 1).
    // Returns true when a hit is committed; for mode 1 `transmission` is the ShadowPayload.
    bool traverse(const RayDesc& r, uint32_t ray_flags, uint32_t mask, int mode, Hit& best, float& transmission) {
        best.t = r.tmax; best.tri = -1;
        if (mask == 0 || o.bvh.tris.empty()) return false;
        bool committed = false, stop = false;
        // 1 / direction clamped to +-1e30: with an infinite reciprocal (a direction component that is exactly zero) an origin that lies ON a box
        // plane gives 0 * inf = NaN and the ray misses a box it is inside of (the HIP traversal clamps the same way, pt_traverse.h trav_init)
        auto rcp = [](float d) { return hmax(hmin(1.0f / d, 1.0e30f), -1.0e30f); };
        float3 inv = {rcp(r.direction.x), rcp(r.direction.y), rcp(r.direction.z)};
        uint64_t nn = 0, nt = 0;
        auto test_tri = [&](int ti) {
            const Tri& t = o.bvh.tris[ti];
            nt++;
            float tt, u, v; bool front;
            // alpha-shadow rays visit every candidate in the ORIGINAL interval (a DXR-conformant
            // far-to-near order; quirk q12), all other rays shrink the interval on commit.
            bool all_candidates = (mode == 1) && (ray_flags & RAY_FLAG_FORCE_NON_OPAQUE);
            if (!intersect(t, r, inv, all_candidates ? r.tmax : best.t, tt, u, v, front, (mode == 0 && best.tri >= 0) ? &o.bvh.tris[best.tri] : nullptr)) return;
            const InstanceDesc& id = o.instances[t.inst];
            if (!(mask & id.instance_mask)) return;
            if (!(id.instance_flags & INSTANCE_FLAG_TRIANGLE_CULL_DISABLE)) {
                if ((ray_flags & RAY_FLAG_CULL_BACK_FACING_TRIANGLES) && !front) return;
                if ((ray_flags & RAY_FLAG_CULL_FRONT_FACING_TRIANGLES) && front) return;
            }
            bool non_opaque = (id.instance_flags & INSTANCE_FLAG_FORCE_NON_OPAQUE) || (ray_flags & RAY_FLAG_FORCE_NON_OPAQUE);
            if (non_opaque) {
                if (mode == 0) { if (!any_hit_alpha_test(t, u, v)) return; }         // IgnoreHit
                else {
                    transmission *= 1 - shadow_alpha(t, u, v);
                    if (transmission == 0.0f) stop = true;                          // AcceptHitAndEndSearch
                }
            }
            committed = true;
            if (!all_candidates || tt < best.t) { best.t = tt; best.u = u; best.v = v; best.tri = ti; best.front = front; }
            if (ray_flags & RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH) stop = true;
        };
        if (o.brute_force || o.bvh.nodes.empty()) {
            for (int i = 0; i < (int)o.bvh.tris.size() && !stop; i++) test_tri(i);
        } else {
            int stack[128]; int sp = 0; int cur = o.bvh.root;
            while (!stop) {
                if (cur < 0) { test_tri(~cur); if (sp == 0) break; cur = stack[--sp]; continue; }
                const BNode& n = o.bvh.nodes[cur];
                nn++;
                float tn[2]; bool hitc[2];
                bool all_candidates = (mode == 1) && (ray_flags & RAY_FLAG_FORCE_NON_OPAQUE);
                float limit = all_candidates ? r.tmax : best.t;
                for (int c = 0; c < 2; c++) {
                    float3 t0 = (n.lo[c] - r.origin) * inv, t1 = (n.hi[c] - r.origin) * inv;
                    float tmn = hmax(hmax(hmin(t0.x, t1.x), hmin(t0.y, t1.y)), hmax(hmin(t0.z, t1.z), r.tmin));
                    float tmx = hmin(hmin(hmax(t0.x, t1.x), hmax(t0.y, t1.y)), hmin(hmax(t0.z, t1.z), limit));
                    tmx *= 1.0000004f;
                    hitc[c] = tmn <= tmx; tn[c] = tmn;
                }
                if (hitc[0] && hitc[1]) {
                    int near = tn[1] < tn[0] ? 1 : 0;
                    if (sp < 127) stack[sp++] = n.child[1 - near];
                    cur = n.child[near];
                } else if (hitc[0]) cur = n.child[0];
                else if (hitc[1]) cur = n.child[1];
                else { if (sp == 0) break; cur = stack[--sp]; }
            }
        }
        t_tally.nodes += nn;
        t_tally.tris += nt;
        if ((int64_t)px == o.log_px && (int64_t)py == o.log_py) {
            const bool got = committed && best.tri >= 0;
            const float rec[16] = {r.origin.x, r.origin.y, r.origin.z, r.tmin, r.direction.x, r.direction.y, r.direction.z, r.tmax, (float)mode, committed ? 1.f : 0.f,
                                   got ? best.t : 0.f, got ? (float)o.bvh.tris[best.tri].inst : -1.f, got ? (float)o.bvh.tris[best.tri].prim : -1.f, transmission, (float)ray_flags, 0.f};
            o.ray_log.insert(o.ray_log.end(), rec, rec + 16);
        }
        return committed;
    }

    float4 rand4(int& count) { return GenerateNextRandom(px, py, sc.seed, count); }

    // TraceShadowRay, :724-742
    float TraceShadowRay(float3 origin, float3 direction, bool alpha_shadow) {
        if (sc.flags & F_INDIRECT_ENVIRONMENT_ONLY) return 1.0f;
        uint32_t rf = (sc.flags & F_CULL_BACKFACE) ? RAY_FLAG_CULL_BACK_FACING_TRIANGLES : 0;
        rf |= RAY_FLAG_SKIP_CLOSEST_HIT_SHADER;
        RayDesc ray = {origin, 0, direction, sc.max_ray_length};
        float transmission = 0.0f;
        if (alpha_shadow) { transmission = 1.0f; rf |= RAY_FLAG_FORCE_NON_OPAQUE; }
        else rf |= RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH;
        t_tally.shadow++;
        Hit h;
        bool hit = traverse(ray, rf, 0xff, 1, h, transmission);
        if (!hit) transmission = 1.0f;                     // ShadowMiss :1081-1085
        return transmission;
    }
    // Miss, :1037-1051
    void Miss(Payload& payload, const RayDesc& ray) {
        if (sc.flags & F_ENVIRONMENT_MAP) {
            payload.color = env ? sc.environment_intensity * SampleCubeLevel(*env, ray.direction, 0) : F3(0);
            if ((sc.flags & F_ENVIRONMENT_MIS) && (payload.flags & PAYLOAD_FLAG_MIS)) {
                float3 l = normalize(ray.direction);
                float env_pdf = env ? ImportanceMapPdf(*env, UnitSquareToUv(SphereToSquare(l))) / (4 * PI) : 0.f;   // :705-710
                payload.color *= BalanceHeuristic(payload.bsdf_pdf, env_pdf);
            }
        } else payload.color = sc.environment_intensity * sc.environment_color;
    }
    void TraceRay(uint32_t ray_flags, uint32_t mask, const RayDesc& ray, Payload& payload) {
        Hit h; float dummy = 0;
        if (traverse(ray, ray_flags, mask, 0, h, dummy)) ClosestHit(payload, ray, h);
        else Miss(payload, ray);
    }
    // TraceBounceRay, :669-678
    float3 TraceBounceRay(float3 origin, float3 direction, int seed, int bounce, float3 throughput, float bsdf_pdf, bool use_mis) {
        const uint32_t mask = (sc.flags & F_INDIRECT_ENVIRONMENT_ONLY) ? 0 : 0xff;
        const uint32_t rf = (sc.flags & F_CULL_BACKFACE) ? RAY_FLAG_CULL_FRONT_FACING_TRIANGLES : 0;   // (sic) quirk q2
        RayDesc ray = {origin, 0, direction, sc.max_ray_length};
        Payload p = {throughput, bsdf_pdf, F3(0), use_mis ? (uint32_t)PAYLOAD_FLAG_MIS : 0u, bounce + 1, seed};
        t_tally.bounce++;
        TraceRay(rf, mask, ray, p);
        return p.color;
    }
    // ClosestHit, :788-1007
    void ClosestHit(Payload& payload, const RayDesc& ray, const Hit& h) {
        t_tally.hits++;
        const Tri& tri = o.bvh.tris[h.tri];
        float3 bw = {1 - h.u - h.v, h.u, h.v};                                     // :150-153
        const Instance& instance = o.instances[tri.inst].gpu;
        const Material& material = o.materials[instance.material_id];
        VertexAttributes va = GetVertexAttributes(o, instance, tri.prim, bw);
        switch (sc.debug_output) {                                               // :806-840
            case DBG_HIT_KIND: payload.color = h.front ? float3{1, 0, 0} : float3{0, 1, 0}; return;
            case DBG_VERTEX_COLOR: payload.color = xyz(va.color); return;
            case DBG_VERTEX_ALPHA: payload.color = F3(va.color.w); return;
            case DBG_VERTEX_NORMAL: payload.color = (va.normal + 1) / 2; return;
            case DBG_VERTEX_TANGENT: payload.color = (xyz(va.tangent) + 1) / 2; return;
            case DBG_VERTEX_BITANGENT: payload.color = (va.bitangent + 1) / 2; return;
            case DBG_TEXCOORD_0: payload.color = {va.texcoords[0].x, va.texcoords[0].y, 0}; return;
            case DBG_TEXCOORD_1: payload.color = {va.texcoords[1].x, va.texcoords[1].y, 0}; return;
            default: break;
        }
        if (!h.front) {                                                          // :842-846
            va.geometric_normal = -va.geometric_normal;
            va.normal = -va.normal;
            va.tangent = -va.tangent;
        }
        float3 intersection = ray.origin + (ray.direction * h.t);                // :849
        float3 ray_origin = OffsetRay(va.position, va.geometric_normal);
        float3 ray_origin_below = OffsetRay(va.position, -va.geometric_normal);
        float3 view = -normalize(ray.direction);
        SurfaceProperties sp = GetSurfaceProperties(o, sc, material, va, view);
        sp.roughness_squared = hmax(sp.roughness_squared, F2(MINIMUM_ROUGHNESS));
        sp.clearcoat_roughness = hmax(sp.clearcoat_roughness, MINIMUM_ROUGHNESS);
        if (sc.flags & F_MATERIAL_USE_GEOMETRIC_NORMALS) { sp.shading_normal = va.geometric_normal; sp.clearcoat_normal = va.geometric_normal; }
        switch (sc.debug_output) {                                               // :863-917
            case DBG_COLOR: payload.color = sp.albedo; return;
            case DBG_ALPHA: payload.color = F3(sp.alpha); return;
            case DBG_SHADING_NORMAL: payload.color = (sp.shading_normal + 1) / 2; return;
            case DBG_SHADING_TANGENT: payload.color = (sp.anisotropy_tangent + 1) / 2; return;
            case DBG_SHADING_BITANGENT: payload.color = (sp.anisotropy_bitangent + 1) / 2; return;
            case DBG_METALNESS: payload.color = F3(sp.metalness); return;
            case DBG_ROUGHNESS: payload.color = F3(sqrtf(sp.roughness_squared.y)); return;
            case DBG_SPECULAR: payload.color = F3(sp.specular_factor); return;
            case DBG_SPECULAR_COLOR: payload.color = sp.specular_color; return;
            case DBG_CLEARCOAT: payload.color = F3(sp.clearcoat); return;
            case DBG_CLEARCOAT_ROUGHNESS: payload.color = F3(sp.clearcoat_roughness); return;
            case DBG_CLEARCOAT_NORMAL: payload.color = (sp.clearcoat_normal + 1) / 2; return;
            case DBG_TRANSMISSIVE: payload.color = F3(sp.transmissive); return;
            default: break;
        }
        if (sc.debug_output == DBG_HEMISPHERE_VIEW_SIDE) {                       // :919-922
            payload.color = dot(view, sp.shading_normal) > 0 ? float3{0, 1, 0} : float3{1, 0, 0};
            return;
        }
        ShadingEnv senv{&o.lut, sc.flags};
        payload.color += GetEmissive(o, material, va.texcoords);                 // :925-926
        if (payload.bounce < sc.max_bounces) {                                   // :929-942
            if ((sc.flags & F_ENVIRONMENT_MAP) && (sc.flags & F_ENVIRONMENT_MIS)) {
                float light_pdf = 0;
                float4 r = rand4(payload.random_count);
                LightRay lr;
                if (env) {                                                       // SampleEnvironmentMap :688-703
                    float2 uv = SampleImportanceMap(*env, {r.x, r.y}, light_pdf);
                    lr.direction = SquareToSphere(UvToUnitSquare(uv));
                    light_pdf /= 4 * PI;
                    lr.color = sc.environment_intensity * SampleCubeLevel(*env, lr.direction, 0);
                } else { lr.direction = {0, 0, 1}; lr.color = F3(0); light_pdf = 1; }
                lr.color *= TraceShadowRay(ray_origin, lr.direction, false);
                if (any_gt0(lr.color)) {
                    float bsdf_pdf = 0;
                    float3 bsdf = EvaluateBsdf(senv, sp, va.geometric_normal, view, lr.direction, bsdf_pdf);
                    float mis = BalanceHeuristic(light_pdf, bsdf_pdf);
                    payload.color += (mis * bsdf * lr.color) / light_pdf;
                }
            }
        }
        if ((sc.flags & F_POINT_LIGHTS) && (sc.num_of_lights > 0)) {             // :945-956
            float u = rand4(payload.random_count).x;
            // SamplePointLight :680-686 (clamp with u possibly 1.0, quirk q17)
            uint32_t li = f2u(u * (float)sc.num_of_lights);
            if (li > (uint32_t)(sc.num_of_lights - 1)) li = (uint32_t)(sc.num_of_lights - 1);
            float pdf = 1.0f / (float)sc.num_of_lights;
            LightRay lr = GetLightRay(o.lights[li], intersection);
            if (sc.flags & F_SHADOW_RAYS) lr.color *= TraceShadowRay(ray_origin, lr.direction, (sc.flags & F_ALPHA_SHADOWS) != 0);
            if (any_gt0(lr.color)) {
                float bsdf_pdf = 0;
                float3 bsdf = EvaluateBsdf(senv, sp, va.geometric_normal, view, lr.direction, bsdf_pdf);
                payload.color += (lr.color * bsdf) / pdf;
            }
        }
        if (payload.bounce < sc.max_bounces) {                                   // :958-1006
            float3 v = view;
            float4 r = rand4(payload.random_count);
            float3 u = {r.x, r.y, r.z};
            bool is_transmission = false, use_mis = false;
            float bsdf_pdf = 1;
            float3 l = F3(0);
            float3 bsdf = SampleBsdf(senv, sp, u, v, l, bsdf_pdf, is_transmission, use_mis);
            float3 weight = bsdf_pdf != 0 ? bsdf / bsdf_pdf : F3(0);
            float3 throughput = payload.throughput * weight;
            switch (sc.debug_output) {
                case DBG_BOUNCE_DIRECTION: payload.color = 0.5f * (l + 1); return;
                case DBG_BOUNCE_BSDF: payload.color = bsdf; return;
                case DBG_BOUNCE_PDF: payload.color = F3(bsdf_pdf); return;
                case DBG_BOUNCE_WEIGHT: payload.color = weight; return;
                case DBG_BOUNCE_IS_TRANSMISSION: payload.color = is_transmission ? float3{0, 1, 0} : float3{1, 0, 0}; return;
                default: break;
            }
            if (any_gt0(throughput)) {
                float ur = rand4(payload.random_count).x;                        // drawn even below min_bounces (quirk q6)
                if (payload.bounce < sc.min_bounces || RussianRoulette(sc.min_rr, sc.max_rr, ur, throughput, weight)) {
                    payload.color += weight * TraceBounceRay(is_transmission ? ray_origin_below : ray_origin, l, payload.random_count,
                                                             payload.bounce, throughput * weight, bsdf_pdf, use_mis);   // weight twice (quirk q5)
                }
            }
        }
    }
    // RayGeneration, :744-786
    void RayGeneration(float* out_rgba) {
        const uint32_t rf = (sc.flags & F_CULL_BACKFACE) ? RAY_FLAG_CULL_BACK_FACING_TRIANGLES : 0;
        Payload payload = {F3(1), 0, F3(0), 0, 0, 0};
        float4 r = rand4(payload.random_count);
        float2 jitter = float2{r.x, r.y} - 0.5f;
        // GenerateCameraRay :131-142
        float2 clip = ((float2{(float)px, (float)py} + 0.5f + jitter) / float2{(float)sc.res_x, (float)sc.res_y}) * 2 - 1;
        clip.y = -clip.y;
        float4 start = mul(sc.clip_to_world, float4{clip.x, clip.y, 1, 1});
        float4 end = mul(sc.clip_to_world, float4{clip.x, clip.y, 0, 1});
        float3 origin = xyz(start) / start.w;
        float3 dir = xyz(end) / end.w - origin;
        RayDesc ray = {origin, 0, normalize(dir), length(dir)};
        t_tally.primary++;
        TraceRay(rf, 0xff, ray, payload);
        if (any_nan(payload.color)) payload.color = (sc.flags & F_SHOW_NAN) ? float3{1, 0, 0} : F3(0);
        if (any_inf(payload.color)) payload.color = (sc.flags & F_SHOW_INF) ? float3{1, 0, 0} : F3(0);
        if (sc.flags & F_LUMINANCE_CLAMP) {
            float lum = Luminance(payload.color);
            if (lum > sc.luminance_clamp) payload.color *= sc.luminance_clamp / lum;
        }
        float4* outp = (float4*)out_rgba + ((size_t)py * sc.res_x + px);
        if ((sc.flags & F_ACCUMULATE) && (sc.accumulated_frames != 0)) {
            float4 history = *outp;
            float blend = 1.0f / ((float)sc.accumulated_frames + 1.0f);
            *outp = lerp(history, F4(payload.color, 1.0f), blend);
        } else *outp = F4(payload.color, 1.0f);
    }
};

// ------------------------------------------------------------------------------------------------
// glm closed forms.  Inverses in fp64, rounded once (difference < 1e-6 rel).
static void mat4_mul(const float* a, const float* b, float* out) {     // column-major a*b
    for (int c = 0; c < 4; c++) for (int r = 0; r < 4; r++) {
        float s = 0; for (int k = 0; k < 4; k++) s += a[k * 4 + r] * b[c * 4 + k];
        out[c * 4 + r] = s;
    }
}
static bool mat4_inverse(const float* mf, float* out) {
    double m[16], inv[16];
    for (int i = 0; i < 16; i++) m[i] = mf[i];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    if (det == 0) return false;
    for (int i = 0; i < 16; i++) out[i] = (float)(inv[i] / det);
    return true;
}

// BuildAllBlas + BuildTlas restated as one world-space triangle soup (Pathtracer.cpp:138-257)
static void build_accel(Oracle& o, int nthreads = 1) {
    auto t0 = std::chrono::steady_clock::now();
    std::vector<size_t> first(o.instances.size() + 1, 0);
    for (size_t ii = 0; ii < o.instances.size(); ii++) first[ii + 1] = first[ii] + o.instances[ii].num_of_indices / 3;
    o.bvh.tris.assign(first.back(), Tri());
    std::atomic<size_t> next_chunk{0};
    const size_t kChunk = 4096;                            // triangles per unit of work: (instance, range) pairs found by a search in `first`
    const size_t n_chunks = (first.back() + kChunk - 1) / kChunk;
    auto fill = [&]() {
        for (;;) {
            const size_t ck = next_chunk.fetch_add(1);
            if (ck >= n_chunks) break;
            const size_t b = ck * kChunk, e = std::min(first.back(), b + kChunk);
            size_t ii = (size_t)(std::upper_bound(first.begin(), first.end(), b) - first.begin()) - 1;
            for (size_t g = b; g < e; ) {
                while (first[ii + 1] <= g) ii++;
                const InstanceDesc& id = o.instances[ii];
                const Instance& in = id.gpu;
                const float* M = in.transform.m;
                double det = (double)M[0] * ((double)M[5] * M[10] - (double)M[9] * M[6]) - (double)M[4] * ((double)M[1] * M[10] - (double)M[9] * M[2]) +
                             (double)M[8] * ((double)M[1] * M[6] - (double)M[5] * M[2]);
                const size_t stop = std::min(e, first[ii + 1]);
                for (; g < stop; g++) {
                    const uint32_t p = (uint32_t)(g - first[ii]);
                    uint32_t v[3];
                    GetIndices(o, in.index_descriptor, p, v);
                    float3 w[3];
                    for (int k = 0; k < 3; k++) w[k] = xyz(mul(in.transform, F4(fetch_pos(o, in.position_descriptor, v[k]), 1)));
                    Tri t; t.v0 = w[0]; t.e1 = w[1] - w[0]; t.e2 = w[2] - w[0]; t.inst = (uint32_t)ii; t.prim = p; t.flags = det < 0 ? 1u : 0u;
                    o.bvh.tris[g] = t;
                }
            }
        }
    };
    if (nthreads <= 1) fill();
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; t++) th.emplace_back(fill);
        for (auto& t : th) t.join();
    }
    if (getenv("ORC_PROFILE")) fprintf(stderr, "  flatten %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    build_lbvh(o.bvh, nthreads);
    o.accel_dirty = false;
    o.last_accel_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// Pathtracer::PathtraceScene, Source/Pathtracer.cpp:259-367
static void pathtrace_scene(Oracle& o, const Settings& s, const ExecuteParams& ep, int nthreads) {
    float world_to_clip[16], clip_to_world[16], view_to_world[16];
    mat4_mul(ep.view_to_clip, ep.world_to_view, world_to_clip);
    mat4_inverse(ep.world_to_view, view_to_world);
    mat4_inverse(world_to_clip, clip_to_world);
    bool reset = memcmp(world_to_clip, o.previous_world_to_clip, 64) != 0 || s.reset;
    if (reset) o.accumulated_frames = 0;
    if (o.accumulated_frames < s.max_accumulated_frames) {
        if (o.accel_dirty) build_accel(o);
        SceneConstants sc;
        memcpy(sc.clip_to_world.m, clip_to_world, 64);
        sc.camera_pos = {view_to_world[12], view_to_world[13], view_to_world[14]};
        sc.num_of_lights = ep.light_count;
        sc.res_x = ep.width; sc.res_y = ep.height;
        sc.seed = s.use_frame_as_seed ? (uint32_t)ep.frame : s.seed;
        sc.accumulated_frames = o.accumulated_frames;
        sc.environment_color = {s.environment_color[0], s.environment_color[1], s.environment_color[2]};
        sc.environment_intensity = s.environment_intensity;
        sc.debug_output = s.debug_output;
        sc.flags = s.flags;
        sc.max_ray_length = 1000;                                               // :322 (setting ignored)
        sc.min_bounces = std::min(std::max(s.min_bounces, 0), o.bounce_limit);
        sc.max_bounces = std::min(std::max(s.max_bounces, 0), o.bounce_limit);
        sc.env = ep.environment_map;
        sc.luminance_clamp = s.luminance_clamp; sc.min_rr = s.min_rr; sc.max_rr = s.max_rr;
        const EnvMap* env = (ep.environment_map >= 0 && ep.environment_map < (int)o.envs.size()) ? &o.envs[ep.environment_map] : nullptr;
        auto t0 = std::chrono::steady_clock::now();
        uint32_t tiles_x = (ep.width + 15) / 16, tiles_y = (ep.height + 15) / 16, ntiles = tiles_x * tiles_y;
        uint32_t rank = ep.tile_rank, nrank = ep.tile_rank_count ? ep.tile_rank_count : 1;
        std::atomic<uint32_t> next{0};
        auto worker = [&]() {
            for (;;) {
                uint32_t t = next.fetch_add(1);
                if (t >= ntiles) break;
                if (t % nrank != rank) continue;
                uint32_t tx = t % tiles_x, ty = t / tiles_x;
                for (uint32_t y = std::max(ty * 16, o.win[1]); y < std::min(std::min(ty * 16 + 16, ep.height), o.win[3]); y++)
                    for (uint32_t x = std::max(tx * 16, o.win[0]); x < std::min(std::min(tx * 16 + 16, ep.width), o.win[2]); x++) {
                        Tracer tr{o, sc, x, y, env};
                        tr.RayGeneration((float*)ep.output);
                    }
            }
            merge_tally(o.counters);
        };
        if (nthreads <= 1) worker();
        else {
            std::vector<std::thread> th;
            for (int i = 0; i < nthreads; i++) th.emplace_back(worker);
            for (auto& t : th) t.join();
        }
        o.last_trace_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (s.flags & F_ACCUMULATE) o.accumulated_frames++;
        else o.accumulated_frames = 0;
    }
    memcpy(o.previous_world_to_clip, world_to_clip, 64);
}

// ------------------------------------------------------------------------------------------------
// Environment preprocessing: ConvertEquirectangularToCubemap.cs.hlsl, GenerateMipLevelArray.cs.hlsl,
// GenerateEnvironmentImportanceMap.cs.hlsl, GenerateEnvironmentImportanceMapLevel.cs.hlsl driven as in
// EnvironmentMap::CreateEnvironmentMap (EnvironmentMap.cpp:84-130,291-346,403-455).
static float3 sample_equirect(const float* rgb, int w, int h, float2 uv) {     // static sampler s1: linear, wrap
    float x = uv.x * (float)w - 0.5f, y = uv.y * (float)h - 0.5f;
    float fx0 = floorf(x), fy0 = floorf(y);
    float fx = x - fx0, fy = y - fy0;
    int i0 = address((int)fx0, w, 0), i1 = address((int)fx0 + 1, w, 0), j0 = address((int)fy0, h, 0), j1 = address((int)fy0 + 1, h, 0);
    auto px = [&](int i, int j) { const float* p = rgb + ((size_t)j * w + i) * 3; return float3{p[0], p[1], p[2]}; };
    float w00 = (1 - fx) * (1 - fy), w10 = fx * (1 - fy), w01 = (1 - fx) * fy, w11 = fx * fy;
    return px(i0, j0) * w00 + px(i1, j0) * w10 + px(i0, j1) * w01 + px(i1, j1) * w11;
}
static void env_build_importance(EnvMap& e) {
    int R = e.imp_res;
    int levels = 0; for (int r = R; r >= 1; r >>= 1) levels++;
    e.imp.assign(levels, {});
    e.imp[0].resize((size_t)R * R);
    int in_size = e.cube_n[0], in_mips = (int)e.cube.size();
    // mip_level = clamp(log2((6*N)/R), 0, mips) with INTEGER division (quirk q10)
    float mip_level = clamp(log2f((float)((6u * (uint32_t)in_size) / (uint32_t)R)), 0.f, (float)in_mips);
    for (int y = 0; y < R; y++) for (int x = 0; x < R; x++) {
        float2 uv = PixelToUV({x, y}, {R, R});
        float3 d = SquareToSphere(UvToUnitSquare(uv));
        e.imp[0][(size_t)y * R + x] = Luminance(SampleCubeLevel(e, d, mip_level));
    }
    for (int l = 1; l < levels; l++) {
        int n = R >> l, pn = R >> (l - 1);
        e.imp[l].resize((size_t)n * n);
        for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) {
            const std::vector<float>& in = e.imp[l - 1];
            float sum = in[(size_t)(2 * y) * pn + 2 * x];
            sum += in[(size_t)(2 * y + 1) * pn + 2 * x];
            sum += in[(size_t)(2 * y) * pn + 2 * x + 1];
            sum += in[(size_t)(2 * y + 1) * pn + 2 * x + 1];
            e.imp[l][(size_t)y * n + x] = sum;
        }
    }
}
static void env_build_cube_mips(EnvMap& e) {
    int N = e.cube_n[0];
    int mips = 1; for (int n = N; n > 1; n >>= 1) mips++;
    e.cube.resize(mips); e.cube_n.resize(mips);
    for (int l = 1; l < mips; l++) {
        int n = N >> l, pn = e.cube_n[l - 1];
        e.cube_n[l] = n;
        e.cube[l].assign((size_t)6 * n * n * 4, 0);
        for (int f = 0; f < 6; f++) for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) {
            float3 r = F3(0);
            r += cube_texel(e.cube[l - 1], pn, f, 2 * x, 2 * y);
            r += cube_texel(e.cube[l - 1], pn, f, 2 * x + 1, 2 * y);
            r += cube_texel(e.cube[l - 1], pn, f, 2 * x, 2 * y + 1);
            r += cube_texel(e.cube[l - 1], pn, f, 2 * x + 1, 2 * y + 1);
            r *= 0.25f;
            uint16_t* p = &e.cube[l][(((size_t)f * n + y) * n + x) * 4];
            p[0] = float_to_half(r.x); p[1] = float_to_half(r.y); p[2] = float_to_half(r.z); p[3] = float_to_half(0.f);
        }
    }
}
static void env_from_equirect(EnvMap& e, const float* rgb, int w, int h) {
    int N = std::max((w / 4) / 2, 1) + 1;                                       // EnvironmentMap.cpp:92 (quirk q11)
    e.N = N;
    e.cube.assign(1, std::vector<uint16_t>((size_t)6 * N * N * 4));
    e.cube_n.assign(1, N);
    for (int f = 0; f < 6; f++) for (int y = 0; y < N; y++) for (int x = 0; x < N; x++) {
        float2 uv = PixelToUV({x, y}, {N, N});
        float3 d = CubemapToDirection(f, uv);                                   // same table as the shader's switch
        float2 eq = {o_atan2(d.y, d.x) / 6.28318530717f, 1 - ((d.z + 1) / 2)};   // equal-area in z (quirk q8)
        float3 c = sample_equirect(rgb, w, h, eq);
        uint16_t* p = &e.cube[0][(((size_t)f * N + y) * N + x) * 4];
        p[0] = float_to_half(c.x); p[1] = float_to_half(c.y); p[2] = float_to_half(c.z); p[3] = float_to_half(1.0f);
    }
    env_build_cube_mips(e);
    env_build_importance(e);
}

// ------------------------------------------------------------------------------------------------
// Skin.cs.hlsl:53-136
struct Bone { float4x4 transform, inverse_transpose; };
struct SkinParams {
    uint32_t num_of_vertices, input_mesh_flags, output_mesh_flags; int input_position, input_tangent_space, input_joint_weight,
        output_position, output_tangent_space, num_of_morph_targets; float morph_weights[4]; int morph_position[4], morph_tangent_space[4]; int use_mfma;
};
// bone_count: StructuredBuffer<Bone> reads beyond the buffer return zeros under D3D12 robust buffer access (Skin.cs.hlsl:93-101 index
// the buffer with the vertex's raw joint ids): an out-of-range joint contributes a zero matrix.
static void skin_run(Oracle& o, const SkinParams& sp, const Bone* bones_in, int bone_count) {
    static const Bone zero_bone = {};
    auto bone_at = [&](uint32_t id) -> const Bone& { return id < (uint32_t)bone_count ? bones_in[id] : zero_bone; };
    const Bone* bones = bones_in;
    uint32_t in_flags = sp.input_mesh_flags;
    if (!bones) in_flags &= (uint32_t)!(1u << 5);          // GpuSkin.cpp:94: `&= !FLAG` clears ALL flags (quirk q19)
    int nt = std::min(sp.num_of_morph_targets, 4);
    for (uint32_t index = 0; index < sp.num_of_vertices; index++) {
        float3 position = ((const float3*)o.buffers[sp.input_position].data.data())[index];
        float3 normal = {0, 0, 0};
        float4 tangent = {0, 0, 0, 1};
        if (in_flags & (1u << 1)) DecodeTangentSpace(UnpackR10G10B10A2(((const uint32_t*)o.buffers[sp.input_tangent_space].data.data())[index]), normal, tangent);
        for (int i = 0; i < nt; i++) {
            float weight = sp.morph_weights[i];
            if (sp.morph_position[i] != -1) position += weight * ((const float3*)o.buffers[sp.morph_position[i]].data.data())[index];
            if (sp.morph_tangent_space[i] != -1) {
                float3 mn; float4 mt;
                DecodeTangentSpace(UnpackR10G10B10A2(((const uint32_t*)o.buffers[sp.morph_tangent_space[i]].data.data())[index]), mn, mt);
                normal += weight * mn;
                tangent = {tangent.x + weight * mt.x, tangent.y + weight * mt.y, tangent.z + weight * mt.z, tangent.w};
            }
        }
        if (in_flags & (1u << 5)) {
            const uint32_t* bw = (const uint32_t*)o.buffers[sp.input_joint_weight].data.data() + (size_t)index * 4;
            uint32_t ids[4]; float w[4];
            for (int i = 0; i < 2; i++) {
                ids[2 * i] = bw[i] & 0xffff; ids[2 * i + 1] = bw[i] >> 16;
                w[2 * i] = (float)(bw[2 + i] & 0xffff) / 65535.0f; w[2 * i + 1] = (float)(bw[2 + i] >> 16) / 65535.0f;
            }
            float3 sp_pos = {0, 0, 0};
            for (int i = 0; i < 4; i++) sp_pos += w[i] * xyz(mul(bone_at(ids[i]).transform, F4(position, 1.f)));
            position = sp_pos;
            if (in_flags & (1u << 1)) {
                float3 sn = {0, 0, 0};
                for (int i = 0; i < 4; i++) sn += w[i] * xyz(mul(bone_at(ids[i]).inverse_transpose, F4(normal, 0.f)));
                normal = sn;
                float3 st = {0, 0, 0};
                for (int i = 0; i < 4; i++) st += w[i] * xyz(mul(bone_at(ids[i]).transform, F4(xyz(tangent), 0.f)));
                tangent = {st.x, st.y, st.z, tangent.w};
            }
        }
        if (sp.output_mesh_flags & 1u) ((float3*)o.buffers[sp.output_position].data.data())[index] = position;
        if (sp.output_mesh_flags & 2u) {
            float3 tn = normalize(xyz(tangent));
            ((uint32_t*)o.buffers[sp.output_tangent_space].data.data())[index] = EncodeTangentSpaceShader(normalize(normal), {tn.x, tn.y, tn.z, tangent.w});
        }
    }
    o.accel_dirty = true;
}

// ------------------------------------------------------------------------------------------------
// ToneMapper.ps.hlsl:30-101 (dither optional; parity is taken before dither, quirk q21)
static float3 AgxCurve(float3 x) {                                              // :30-44
    float3 x2 = x * x, x4 = x2 * x2;
    float3 r = 15.5f * x4 * x2;
    r = r - 40.14f * x4 * x;
    r = r + 31.96f * x4;
    r = r - 6.868f * x2 * x;
    r = r + 0.4298f * x2;
    r = r + 0.1191f * x;
    r = r - 0.00232f;
    return r;
}
static float3 AgxTonemap(float3 c) {                                            // :49-75
    const float3x3 inset = transpose(M3({0.856627153315983f, 0.137318972929847f, 0.11189821299995f},
                                        {0.0951212405381588f, 0.761241990602591f, 0.0767994186031903f},
                                        {0.0482516061458583f, 0.101439036467562f, 0.811302368396859f}));
    c = mul(inset, c);
    const float log_min = -12.47393f, log_max = 4.026069f;
    c = clamp(float3{o_log2(c.x), o_log2(c.y), o_log2(c.z)}, log_min, log_max);
    c = (c - log_min) / (log_max - log_min);
    c = AgxCurve(c);
    const float3x3 outset = transpose(M3({1.12710058f, -0.14132976f, -0.14132976f}, {-0.11060664f, 1.1578237f, -0.11060664f},
                                         {-0.01649394f, -0.01649394f, 1.25193641f}));
    c = mul(outset, c);
    return hpow(c, 2.2f);
}
struct TonemapConfig { int tonemapper; float exposure; int frame; int dither; };
static float3 tonemap_pixel(const TonemapConfig& cfg, float3 c, uint32_t px, uint32_t py) {
    c = cfg.exposure * c;
    if (cfg.tonemapper == 0) c = saturate(c);
    else c = AgxTonemap(c);
    c = EncodeSrgb(c);
    if (cfg.dither) {                                                           // :77-81
        uint32_t a[3] = {px * 2, py * 2, (uint32_t)cfg.frame * 2}, b[3] = {px * 2 + 1, py * 2 + 1, (uint32_t)cfg.frame * 2 + 1};
        pcg3d(a); pcg3d(b);
        const float d = 4294967296.0f;                                           // float(0xffffffff)
        float3 n = float3{(float)a[0] / d, (float)a[1] / d, (float)a[2] / d} + float3{(float)b[0] / d, (float)b[1] / d, (float)b[2] / d} - 1.0f;
        c = c + n / 255.f;
    }
    return c;
}

}  // namespace orc

// ================================================================================================
// C API (ctypes)
using namespace orc;
extern "C" {

void* orc_create(const float* sheen_e_16x16) {
    init_srgb();
    Oracle* o = new Oracle();
    memcpy(o->lut.v, sheen_e_16x16, sizeof(o->lut.v));
    o->samplers.push_back({0, 0, 1, 1});                  // default sampler 0: linear / wrap (GpuResources.cpp:47-59)
    return o;
}
void orc_destroy(void* h) { delete (Oracle*)h; }
int orc_buffer_create(void* h, const void* data, size_t bytes, int format) {
    Oracle* o = (Oracle*)h;
    Buffer b; b.format = format; b.data.resize(bytes + 16);
    if (data) memcpy(b.data.data(), data, bytes);
    o->buffers.push_back(std::move(b));
    return (int)o->buffers.size() - 1;
}
void orc_buffer_update(void* h, int handle, const void* data, size_t bytes) { Oracle* o = (Oracle*)h; memcpy(o->buffers[handle].data.data(), data, bytes); o->accel_dirty = true; }
void orc_buffer_read(void* h, int handle, void* data, size_t bytes) { memcpy(data, ((Oracle*)h)->buffers[handle].data.data(), bytes); }
int orc_texture_create(void* h, const uint8_t* rgba8, int w, int hh, int srgb) {
    Oracle* o = (Oracle*)h;
    Texture t; t.w = w; t.h = hh; t.srgb = srgb != 0; t.px.assign(rgba8, rgba8 + (size_t)w * hh * 4);
    o->textures.push_back(std::move(t));
    return (int)o->textures.size() - 1;
}
int orc_sampler_create(void* h, const int* desc4) { Oracle* o = (Oracle*)h; o->samplers.push_back({desc4[0], desc4[1], desc4[2], desc4[3]}); return (int)o->samplers.size() - 1; }
void orc_set_materials(void* h, const void* m, int n) { Oracle* o = (Oracle*)h; o->materials.assign((const Material*)m, (const Material*)m + n); }
void orc_set_lights(void* h, const void* l, int n) { Oracle* o = (Oracle*)h; o->lights.assign((const Light*)l, (const Light*)l + n); }
void orc_set_instances(void* h, const void* in, int n) { Oracle* o = (Oracle*)h; o->instances.assign((const InstanceDesc*)in, (const InstanceDesc*)in + n); o->accel_dirty = true; }
int orc_env_create(void* h, const float* rgb, int w, int hh) {
    Oracle* o = (Oracle*)h; o->envs.emplace_back(); env_from_equirect(o->envs.back(), rgb, w, hh); return (int)o->envs.size() - 1;
}
// Tracer-only parity: load maps preprocessed elsewhere (cube mip 0 + the whole pyramid).
int orc_env_create_raw(void* h, int N, const uint16_t* cube_rgba16f, const float* pyramid) {
    Oracle* o = (Oracle*)h; o->envs.emplace_back(); EnvMap& e = o->envs.back();
    e.N = N; e.cube.assign(1, std::vector<uint16_t>(cube_rgba16f, cube_rgba16f + (size_t)6 * N * N * 4)); e.cube_n.assign(1, N);
    env_build_cube_mips(e);
    int R = e.imp_res; const float* p = pyramid;
    for (int r = R; r >= 1; r >>= 1) { e.imp.emplace_back(p, p + (size_t)r * r); p += (size_t)r * r; }
    return (int)o->envs.size() - 1;
}
int orc_env_cube_size(void* h, int env) { return ((Oracle*)h)->envs[env].N; }
void orc_env_read(void* h, int env, uint16_t* cube_rgba16f, float* pyramid) {
    EnvMap& e = ((Oracle*)h)->envs[env];
    if (cube_rgba16f) memcpy(cube_rgba16f, e.cube[0].data(), e.cube[0].size() * 2);
    if (pyramid) for (auto& l : e.imp) { memcpy(pyramid, l.data(), l.size() * 4); pyramid += l.size(); }
}
// the oracle's own math routines on arrays (same op numbers as the HIP library's pt_debug_math test hook)
void orc_math(int op, const float* a, const float* b, float* out, int n) {
    for (int i = 0; i < n; i++) {
        const float x = a[i], y = b[i];
        float r = 0;
        switch (op) {
            case 0: r = o_atan2(x, y); break;
            case 1: r = hpow(x, y); break;
            case 2: r = o_exp(x); break;
            case 3: r = o_log2(x); break;
            case 4: r = o_exp2(x); break;
            case 5: r = o_sin(x); break;
            case 6: r = o_cos(x); break;
            case 7: r = x / y; break;
            case 8: r = hpow5(x); break;
        }
        out[i] = r;
    }
}
void orc_set_bounce_limit(void* h, int limit) { ((Oracle*)h)->bounce_limit = limit; }
void orc_set_brute_force(void* h, int on) { ((Oracle*)h)->brute_force = on != 0; }
// diagnostics: restrict orc_trace to a pixel window (x1 = 0: whole frame); log the rays of one pixel (x < 0: off) and read the log back
void orc_set_window(void* h, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1) {
    Oracle* o = (Oracle*)h;
    if (x1 == 0) { o->win[0] = o->win[1] = 0; o->win[2] = o->win[3] = 0xffffffffu; }
    else { o->win[0] = x0; o->win[1] = y0; o->win[2] = x1; o->win[3] = y1; }
}
void orc_set_ray_log(void* h, int x, int y) { Oracle* o = (Oracle*)h; o->log_px = x; o->log_py = y; o->ray_log.clear(); }
int orc_read_ray_log(void* h, float* out, int max_rays) {
    Oracle* o = (Oracle*)h;
    const int n = (int)std::min<size_t>(o->ray_log.size() / 16, (size_t)std::max(max_rays, 0));
    if (out && n) memcpy(out, o->ray_log.data(), (size_t)n * 16 * sizeof(float));
    return (int)(o->ray_log.size() / 16);
}
void orc_build_accel(void* h) { build_accel(*(Oracle*)h); }
// the same tree built on `nthreads` cores (bench.py cpu_baseline leg B2); 1 = the build above
void orc_build_accel_mt(void* h, int nthreads) { build_accel(*(Oracle*)h, nthreads); }
void orc_skin_run(void* h, const void* params, const void* bones, int bone_count) { skin_run(*(Oracle*)h, *(const SkinParams*)params, (const Bone*)bones, bone_count); }
void orc_trace(void* h, const void* settings, const void* params, int nthreads) { pathtrace_scene(*(Oracle*)h, *(const Settings*)settings, *(const ExecuteParams*)params, nthreads); }
// out[0..6] = primary, bounce, shadow, nodes, tris, closest hits, texture taps; out[7] = accumulated_frames
void orc_get_counters(void* h, uint64_t* out, int reset) {
    Oracle* o = (Oracle*)h; Counters& c = o->counters;
    out[0] = c.primary; out[1] = c.bounce; out[2] = c.shadow; out[3] = c.nodes; out[4] = c.tris; out[5] = c.hits; out[6] = c.taps; out[7] = (uint64_t)o->accumulated_frames;
    if (reset) { c.primary = 0; c.bounce = 0; c.shadow = 0; c.nodes = 0; c.tris = 0; c.hits = 0; c.taps = 0; }
}
void orc_get_timing(void* h, double* accel_ms, double* trace_ms) { Oracle* o = (Oracle*)h; *accel_ms = o->last_accel_ms; *trace_ms = o->last_trace_ms; }
void orc_bvh_info(void* h, uint32_t* nodes, uint32_t* tris) { Oracle* o = (Oracle*)h; *nodes = (uint32_t)o->bvh.nodes.size(); *tris = (uint32_t)o->bvh.tris.size(); }
void orc_tonemap(const void* cfg, const float* rgba, uint32_t w, uint32_t hh, float* out_rgb, uint8_t* out_rgba8) {
    const TonemapConfig& c = *(const TonemapConfig*)cfg;
    for (uint32_t y = 0; y < hh; y++) for (uint32_t x = 0; x < w; x++) {
        // ToneMapper.ps.hlsl:87-88: pixel = UVToPixel(uv, resolution) with uv at the pixel centre
        int2 p = UVToPixel({((float)x + 0.5f) / (float)w, ((float)y + 0.5f) / (float)hh}, {(int)w, (int)hh});
        const float* s = rgba + ((size_t)p.y * w + p.x) * 4;
        float3 t = tonemap_pixel(c, {s[0], s[1], s[2]}, (uint32_t)p.x, (uint32_t)p.y);
        size_t i = (size_t)y * w + x;
        if (out_rgb) { out_rgb[i * 3] = t.x; out_rgb[i * 3 + 1] = t.y; out_rgb[i * 3 + 2] = t.z; }
        if (out_rgba8) {
            float3 q = saturate(t) * 255.f + 0.5f;
            out_rgba8[i * 4] = (uint8_t)q.x; out_rgba8[i * 4 + 1] = (uint8_t)q.y; out_rgba8[i * 4 + 2] = (uint8_t)q.z; out_rgba8[i * 4 + 3] = 255;
        }
    }
}

// ---- unit-level entry points for known-answer tests ---------------------------------------------
void orc_pcg4d(const uint32_t* in4, uint32_t* out4) { uint4 r = pcg4d({in4[0], in4[1], in4[2], in4[3]}); out4[0] = r.x; out4[1] = r.y; out4[2] = r.z; out4[3] = r.w; }
void orc_random(uint32_t px, uint32_t py, uint32_t seed, int count, float* out4) { float4 r = GenerateNextRandom(px, py, seed, count); out4[0] = r.x; out4[1] = r.y; out4[2] = r.z; out4[3] = r.w; }
void orc_octa_encode(const float* n, float* e) { float2 r = EncodeOctahedralMap({n[0], n[1], n[2]}); e[0] = r.x; e[1] = r.y; }
void orc_octa_decode(const float* e, float* n) { float3 r = DecodeOctahedralMap({e[0], e[1]}); n[0] = r.x; n[1] = r.y; n[2] = r.z; }
uint32_t orc_encode_tangent_space_host(const float* n, const float* t4) { return EncodeTangentSpaceHost({n[0], n[1], n[2]}, {t4[0], t4[1], t4[2], t4[3]}); }
uint32_t orc_encode_tangent_space_shader(const float* n, const float* t4) { return EncodeTangentSpaceShader({n[0], n[1], n[2]}, {t4[0], t4[1], t4[2], t4[3]}); }
uint32_t orc_encode_normal_host(const float* n) { return EncodeNormalHost({n[0], n[1], n[2]}); }
void orc_decode_tangent_space(uint32_t packed, float* n3, float* t4) {
    float3 n; float4 t; DecodeTangentSpace(UnpackR10G10B10A2(packed), n, t);
    n3[0] = n.x; n3[1] = n.y; n3[2] = n.z; t4[0] = t.x; t4[1] = t.y; t4[2] = t.z; t4[3] = t.w;
}
void orc_square_to_sphere(const float* s, float* d) { float3 r = SquareToSphere({s[0], s[1]}); d[0] = r.x; d[1] = r.y; d[2] = r.z; }
void orc_sphere_to_square(const float* d, float* s) { float2 r = SphereToSquare({d[0], d[1], d[2]}); s[0] = r.x; s[1] = r.y; }
void orc_uv_to_square(const float* uv, float* s) { float2 r = UvToUnitSquare({uv[0], uv[1]}); s[0] = r.x; s[1] = r.y; }
void orc_square_to_uv(const float* s, float* uv) { float2 r = UnitSquareToUv({s[0], s[1]}); uv[0] = r.x; uv[1] = r.y; }
void orc_square_to_disk(const float* s, float* d) { float2 r = SquareToDisk2({s[0], s[1]}); d[0] = r.x; d[1] = r.y; }
void orc_cubemap_to_direction(int face, const float* uv, float* d) { float3 r = CubemapToDirection(face, {uv[0], uv[1]}); d[0] = r.x; d[1] = r.y; d[2] = r.z; }
void orc_dir_to_face(const float* d, int* face, float* uv) { dir_to_face({d[0], d[1], d[2]}, *face, uv[0], uv[1]); }
void orc_offset_ray(const float* p, const float* n, float* out) { float3 r = OffsetRay({p[0], p[1], p[2]}, {n[0], n[1], n[2]}); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
void orc_light_ray(const void* light, const float* p, float* dir_color6) {
    LightRay r = GetLightRay(*(const Light*)light, {p[0], p[1], p[2]});
    dir_color6[0] = r.direction.x; dir_color6[1] = r.direction.y; dir_color6[2] = r.direction.z; dir_color6[3] = r.color.x; dir_color6[4] = r.color.y; dir_color6[5] = r.color.z;
}
float orc_sheen_e(void* h, float alpha, float cos_theta) { return SheenE(((Oracle*)h)->lut, alpha, cos_theta); }
// sp36 = SurfaceProperties as 36 floats.  out = {bsdf.rgb, pdf}
void orc_evaluate_bsdf(void* h, uint32_t flags, const float* sp36, const float* ng, const float* v, const float* l, float* out4) {
    ShadingEnv env{&((Oracle*)h)->lut, flags};
    SurfaceProperties sp; memcpy(&sp, sp36, sizeof(sp));
    float pdf = 0;
    float3 b = EvaluateBsdf(env, sp, {ng[0], ng[1], ng[2]}, {v[0], v[1], v[2]}, {l[0], l[1], l[2]}, pdf);
    out4[0] = b.x; out4[1] = b.y; out4[2] = b.z; out4[3] = pdf;
}
// out = {bsdf.rgb, pdf, l.xyz, is_transmission, use_mis}
void orc_sample_bsdf(void* h, uint32_t flags, const float* sp36, const float* u3, const float* v, float* out9) {
    ShadingEnv env{&((Oracle*)h)->lut, flags};
    SurfaceProperties sp; memcpy(&sp, sp36, sizeof(sp));
    float3 l; float pdf; bool it, um;
    float3 b = SampleBsdf(env, sp, {u3[0], u3[1], u3[2]}, {v[0], v[1], v[2]}, l, pdf, it, um);
    out9[0] = b.x; out9[1] = b.y; out9[2] = b.z; out9[3] = pdf; out9[4] = l.x; out9[5] = l.y; out9[6] = l.z; out9[7] = it ? 1.f : 0.f; out9[8] = um ? 1.f : 0.f;
}
void orc_sample_texture(void* h, int tex, int sampler, const float* uv, float* out4) {
    Oracle* o = (Oracle*)h; float4 r = SampleLevel0(o->textures[tex], o->samplers[sampler], {uv[0], uv[1]}); out4[0] = r.x; out4[1] = r.y; out4[2] = r.z; out4[3] = r.w;
}
void orc_sample_cube(void* h, int env, const float* d, float level, float* out3) {
    float3 r = SampleCubeLevel(((Oracle*)h)->envs[env], {d[0], d[1], d[2]}, level); out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}
void orc_sample_importance_map(void* h, int env, const float* u2, float* uv_pdf3) {
    float pdf; float2 uv = SampleImportanceMap(((Oracle*)h)->envs[env], {u2[0], u2[1]}, pdf); uv_pdf3[0] = uv.x; uv_pdf3[1] = uv.y; uv_pdf3[2] = pdf;
}
float orc_importance_map_pdf(void* h, int env, const float* uv) { return ImportanceMapPdf(((Oracle*)h)->envs[env], {uv[0], uv[1]}); }
void orc_tonemap_pixel(const void* cfg, const float* rgb, float* out) { float3 r = tonemap_pixel(*(const TonemapConfig*)cfg, {rgb[0], rgb[1], rgb[2]}, 0, 0); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
uint16_t orc_float_to_half(float f) { return float_to_half(f); }
float orc_half_to_float(uint16_t hh) { return half_to_float(hh); }
void orc_mat4_inverse(const float* m, float* out) { mat4_inverse(m, out); }
// closest-hit query for traversal tests: out = {hit, t, u, v, instance, primitive, front}
void orc_intersect(void* h, const float* origin, const float* dir, float tmin, float tmax, uint32_t ray_flags, float* out7) {
    Oracle* o = (Oracle*)h;
    if (o->accel_dirty) build_accel(*o);
    SceneConstants sc{}; Tracer tr{*o, sc, 0xffffffffu, 0xffffffffu, nullptr};
    RayDesc r = {{origin[0], origin[1], origin[2]}, tmin, {dir[0], dir[1], dir[2]}, tmax};
    Hit hit; float dummy = 0;
    bool got = tr.traverse(r, ray_flags, 0xff, 0, hit, dummy);
    merge_tally(o->counters);
    out7[0] = got ? 1.f : 0.f;
    if (got) { const Tri& t = o->bvh.tris[hit.tri]; out7[1] = hit.t; out7[2] = hit.u; out7[3] = hit.v; out7[4] = (float)t.inst; out7[5] = (float)t.prim; out7[6] = hit.front ? 1.f : 0.f; }
}
// many rays at once on `nthreads` cores (rays: 8 floats each, origin tmin direction tmax; out: 8 floats each, committed t u v instance
// primitive front transmission); mode 0 = TraceRay's search, 1 = TraceShadowRay's
void orc_intersect_many(void* h, const float* rays, int n, uint32_t ray_flags, int mode, float* out, int nthreads) {
    Oracle* o = (Oracle*)h;
    if (o->accel_dirty) build_accel(*o);
    std::atomic<int> next{0};
    auto worker = [&]() {
        SceneConstants sc{}; Tracer tr{*o, sc, 0xffffffffu, 0xffffffffu, nullptr};
        for (;;) {
            const int b = next.fetch_add(1024);
            if (b >= n) break;
            for (int i = b; i < std::min(n, b + 1024); i++) {
                const float* q = rays + (size_t)i * 8;
                RayDesc r = {{q[0], q[1], q[2]}, q[3], {q[4], q[5], q[6]}, q[7]};
                Hit hit; float transmission = (mode == 1 && (ray_flags & RAY_FLAG_FORCE_NON_OPAQUE)) ? 1.0f : 0.0f;
                const bool got = tr.traverse(r, ray_flags, 0xff, mode, hit, transmission);
                const bool have = got && hit.tri >= 0;
                float* w = out + (size_t)i * 8;
                w[0] = got ? 1.f : 0.f; w[1] = have ? hit.t : 0.f; w[2] = have ? hit.u : 0.f; w[3] = have ? hit.v : 0.f;
                w[4] = have ? (float)o->bvh.tris[hit.tri].inst : -1.f; w[5] = have ? (float)o->bvh.tris[hit.tri].prim : -1.f; w[6] = (have && hit.front) ? 1.f : 0.f;
                w[7] = transmission;
            }
        }
        merge_tally(o->counters);
    };
    if (nthreads <= 1) worker();
    else { std::vector<std::thread> th; for (int t = 0; t < nthreads; t++) th.emplace_back(worker); for (auto& t : th) t.join(); }
}
}
