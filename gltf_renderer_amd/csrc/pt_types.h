// pt_types.h -- device-side scene layout in HBM for the path-tracing kernels.
//
// Host PODs come from include/mipt.h (byte-identical to the reference's GPU structs).  What the
// reference reaches through bindless descriptor indices (ResourceDescriptorHeap[i]) is resolved on the
// host when a scene is set: instance rows carry their stream pointers (InstanceRec) and materials become
// RMat records with texel pointers, packed sampler state and pre-multiplied UV transforms.  The
// buffer / texture / sampler tables (BufferRec, TextureRec, SamplerRec) live on the host; only the buffer
// table is also copied to the device, for the BVH build.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mipt.h"

namespace pt {

struct BufferRec {          // one vertex / index stream
    const void* ptr;
    uint32_t format;        // pt_format
    uint32_t bytes;
};
struct TextureRec {         // RGBA8, one mip (Gltf.cpp:1059-1060)
    const uint32_t* texels;
    int32_t width, height;
    uint32_t srgb, _pad;
};
struct SamplerRec { int32_t address_u, address_v, min_filter, mag_filter; };

// Instance table row: Instance (PathTracer.lib.hlsl:32-41) + what the TLAS instance desc carried + the stream
// pointers its descriptors resolve to (one dependent load fewer per vertex fetch than going through the table).
struct __attribute__((aligned(16))) InstanceRec {
    pt_mesh_instance gpu;   // 156 B
    uint32_t mask_flags;    // bits 0-7 instance mask, bit 8 cull-disable, bit 9 force-non-opaque, bit 10 mirrored
    uint32_t tri_offset;    // first triangle of this instance in build order
    uint32_t tri_count;
    uint32_t index_is16;    // index stream format (R16_UINT vs R32_UINT)
    uint32_t _pad;
    const void* p_index;            // nullptr = non-indexed
    const float* p_position;
    const uint32_t* p_tangent_space;   // nullptr = absent
    const float2* p_texcoord[2];
    const uint2* p_color;
    uint64_t _pad2;
};
static_assert(sizeof(InstanceRec) == 240, "InstanceRec");

// Resolved material: what the kernels read instead of pt_material + texture/sampler tables.  Built on the host by
// pt_scene_set_materials; laid out so the header is 8 dwordx4 loads issued together and a texture slot is 3.
// The UV transform rows (T*(R*S), Material.hlsli:68-88) are pre-multiplied in fp32 exactly as the shader would.
struct __attribute__((aligned(16))) RTex {
    const uint32_t* texels;        // unbound slots point at a 1x1 white texel, so address math needs no branch
    int32_t width, height;
    uint32_t flags;                // bit 0 sRGB, bits 1-2 address_u, bits 3-4 address_v, bit 5 point filter, bit 6 tex_coord set
    float m00, m01, ox;            // tu = m00*u + m01*v + ox
    float m10, m11, oy;            // tv = m10*u + m11*v + oy
    uint32_t _pad;
};
static_assert(sizeof(RTex) == 48, "RTex");
enum { RT_SRGB = 1, RT_POINT = 32, RT_TEXCOORD1 = 64 };
// bound_mask bits beyond the 15 slots.  RM_TRIO: the albedo slot is bound and the bound ones of the normal and metal-rough slots share
// its image size, sampler state, UV set and UV transform, so the three bilinear footprints are the SAME four texels and `trio` holds
// them side by side: {albedo, normal, metal-rough, emissive} per texel.  One footprint then costs two 32-B pieces instead of six 8-B pieces in
// six different cache lines (a hit's texel fetches were 6 of its ~13 HBM lines).  RM_TRIO_SRGB_N / _M: sRGB flag of the other two.
// RM_TRIO_EMISSIVE: the emissive texture has that footprint too and rides in the copy's fourth component (RM_TRIO_SRGB_E its sRGB flag):
// the one texture fetch a hit made after the batch -- a dependent round trip and two more cache lines -- comes with it.
enum : uint32_t { RM_TRIO = 1u << 16, RM_TRIO_SRGB_N = 1u << 17, RM_TRIO_SRGB_M = 1u << 18, RM_TRIO_EMISSIVE = 1u << 19, RM_TRIO_SRGB_E = 1u << 20 };
enum { SLOT_NORMAL = 0, SLOT_ALBEDO, SLOT_METALLIC_ROUGHNESS, SLOT_OCCLUSION, SLOT_EMISSIVE, SLOT_SPECULAR, SLOT_SPECULAR_COLOR, SLOT_CLEARCOAT,
       SLOT_CLEARCOAT_ROUGHNESS, SLOT_CLEARCOAT_NORMAL, SLOT_ANISOTROPY, SLOT_SHEEN_COLOR, SLOT_SHEEN_ROUGHNESS, SLOT_TRANSMISSION, SLOT_THICKNESS,
       SLOT_COUNT };
struct __attribute__((aligned(16))) RMat {
    uint32_t flags; int32_t alpha_mode; float metalness_factor, roughness_factor;
    float base_color_factor[4];
    float emissive_factor[3], alpha_cutoff;
    float ior, normal_scale, specular_factor, clearcoat_normal_scale;
    float specular_color_factor[3], clearcoat_factor;
    float clearcoat_roughness_factor, anisotropy_strength, anisotropy_cos, anisotropy_sin;   // cos/sin(anisotropy_rotation), fp32
    float sheen_color_factor[3], sheen_roughness_factor;
    float transmission_factor; uint32_t bound_mask;                                          // bit k: slot k has a texture; RM_TRIO* bits below
    const uint4* trio;              // RM_TRIO: albedo / normal / metal-rough texels interleaved, 16 B a texel (pt_scene_set_materials)
    RTex tex[SLOT_COUNT];
};
static_assert(sizeof(RMat) == 128 + 48 * 15, "RMat");

// 64-B BVH2 node: both children's boxes + child references (>=0 inner node, <0 leaf: ~triangle index).
struct __attribute__((aligned(64))) BvhNode {
    float lo0[3], hi0[3], lo1[3], hi1[3];
    int32_t child0, child1;
    uint32_t _pad[2];
};
static_assert(sizeof(BvhNode) == 64, "BvhNode");

// 64-B 4-wide node, the children's boxes quantised to 8 bits per plane on the grid origin + q * 2^(exp - 127) spanned by the
// node's own box (three dwordx4 + one dwordx2 load per node step instead of seven dwordx4: the traversal stages are bound by
// vector-memory instruction issue, not by VALU, so the dequantisation is free and the loads saved are time saved).
// Quantisation is conservative (lo rounded down, hi rounded up, checked with the very fma the traversal uses), so a ray enters a
// superset of the children it would enter with exact boxes and finds the same hits.
// child >= 0: wide node index; child < 0: leaf reference (below); kEmptyChild: unused slot (never entered).
#ifndef PT_BVH_WIDTH
#define PT_BVH_WIDTH 4          // children per wide node: 4 (64-B node, 4 loads per step) or 8 (96-B node, 6 loads per step).  Measured:
                                // 8-wide takes 10.4 node steps per ray instead of 15.4 with the same loads per ray and is 7 % SLOWER
                                // (4200 against 4530 Mrays/s: eight dequantised slab tests per step are no longer free)
#endif
constexpr int kBvhWidth = PT_BVH_WIDTH;
static_assert(kBvhWidth == 4 || kBvhWidth == 8, "PT_BVH_WIDTH");
#if PT_BVH_WIDTH == 4
struct __attribute__((aligned(64))) Bvh4Node {
    float origin[3];                // lo corner of the union of the children
    uint8_t exp[3], _e;             // biased exponents of the per-axis grid step
    int32_t child[4];
    uint8_t qlox[4], qhix[4], qloy[4], qhiy[4];     // byte k of each word = child k (v_cvt_f32_ubyte<k>)
    uint8_t qloz[4], qhiz[4];
    uint32_t _pad[2];
};
static_assert(sizeof(Bvh4Node) == 64, "Bvh4Node");
#else
// 8-wide: the same encoding with eight children, 96 B = six dwordx4 loads.  A ray takes about half as many node steps, and a
// step is one dependent round trip to memory whatever it fetches.
struct __attribute__((aligned(16))) Bvh4Node {
    float origin[3];
    uint8_t exp[3], _e;
    int32_t child[8];
    uint8_t qlox[8], qhix[8], qloy[8], qhiy[8], qloz[8], qhiz[8];     // byte k & 3 of word k >> 2 of each bound = child k
};
static_assert(sizeof(Bvh4Node) == 96, "Bvh4Node");
#endif
constexpr int kNodeFloat4 = (int)(sizeof(Bvh4Node) / 16);
// Per wide node, the contiguous range [first, first + count) of Morton-sorted triangles under each child (count 0: unused slot).
// Written by the collapse, read by the refit: a child's box is a range query over the triangles' boxes.
struct __attribute__((aligned(16))) WideRanges { uint32_t first[PT_BVH_WIDTH], count[PT_BVH_WIDTH]; };
constexpr int32_t kEmptyChild = 0x7fffffff;
// the plane a quantised coordinate stands for; build and traversal must use this one expression
__host__ __device__ __forceinline__ float bvh_dequant(uint32_t q, float step, float origin) { return __builtin_fmaf((float)q, step, origin); }
__host__ __device__ __forceinline__ float bvh_step(uint32_t biased_exp) {
    union { uint32_t u; float f; } c; c.u = biased_exp << 23; return c.f;
}
// Leaf reference: ~(first | (count - 1) << 28): `count` (1..kLeafMax) triangle packets starting at `first`, contiguous because
// an LBVH subtree covers a contiguous range of the Morton-sorted triangles.  first < 2^28.
#ifndef PT_LEAF_MAX
#define PT_LEAF_MAX 3           // 1 = one triangle per leaf reference; measured 1: 3423, 2: 3508, 3: 3513, 4: 3414 Mrays/s
#endif
constexpr int kLeafMax = PT_LEAF_MAX;
constexpr uint32_t kLeafFirstMask = 0x0fffffffu;

// 48-B triangle packet in world space: v0, e1 = v1-v0, e2 = v2-v0 + ids -- stored one per 64-B line (PT_TRI_STRIDE64 = 1): packed at 48 B a
// packet straddles two lines a third of the time, and a leaf test then waits for two cache lines; one per line measured 13.9 against 14.05 ms
// of traversal per launch (+1 % rays/s) for 4 MB more on the bench scene (257 k triangles: 16.4 instead of 12.3 MB).  Bit-identical images.
#ifndef PT_TRI_STRIDE64
#define PT_TRI_STRIDE64 1
#endif
struct __attribute__((aligned(16))) TriPacket {
    float v0[3]; uint32_t inst;
    float e1[3]; uint32_t prim;
    float e2[3]; uint32_t flags;   // copy of InstanceRec::mask_flags
#if PT_TRI_STRIDE64
    uint32_t _line_pad[4];
#endif
};
static_assert(sizeof(TriPacket) == (PT_TRI_STRIDE64 ? 64 : 48), "TriPacket");
constexpr int kTriFloat4 = (int)(sizeof(TriPacket) / 16);

// 128-B shading packet of one triangle, same (Morton) order as the TriPacket array: everything GetVertexAttributes gathers for the
// three vertices, de-indexed at build time, so a hit reads ONE cache line instead of an index triple plus 9-15 scattered
// 64-B sectors of the per-attribute streams (the shade stage moves ~4 TB/s of HBM traffic; this is a quarter of it).
// Absent streams hold zeros; which streams exist is still told by the instance row's stream pointers.
// The first 80 B (five dwordx4) hold what EVERY hit reads -- positions, packed tangent spaces, the first UV set, the instance id; the second
// UV set and the vertex colours follow and are fetched only for meshes that have them (a dependent fetch for those meshes alone).  The
// shade stage is bound by the number of divergent vector-memory instructions it issues (64 different lines each), not by their bytes:
// three fewer per hit.
struct __attribute__((aligned(128))) ShadePacket {
    struct V { float pos[3]; uint32_t tangent_space; } v[3];   // 48 B (object space)
    float uv0[3][2];                // 24 B
    uint32_t inst, _pad;            // instance-table row (copy of TriPacket::inst: the shade stage never touches the TriPacket)
    float uv1[3][2];                // 24 B  -- rarely present from here on
    uint32_t color[3][2];           // 24 B
};
static_assert(sizeof(ShadePacket) == 128, "ShadePacket");

enum : uint32_t { TF_CULL_DISABLE = 1u << 8, TF_FORCE_NON_OPAQUE = 1u << 9, TF_MIRRORED = 1u << 10 };

struct EnvRec {
    const uint16_t* cube;       // mip 0, RGBA16F, [face][y][x][4]
    int32_t cube_n;
    const float* importance;    // sum pyramid, level 0 first
    uint32_t level_offset[12];  // float offset of each level
    int32_t imp_res;            // 1024
    int32_t imp_levels;         // 11
    float imp_total;            // importance[level 10]: the sum of the map (the pdf's normalisation)
    // Copies of levels 1024^2, 256^2, 64^2, 16^2, 4^2 with every 4x4 texel block contiguous (64 B): one cache line feeds two
    // levels of the sampling descent (the in-between level is re-summed from it in the build's order, bit-identically).
    const float* blocked;
    uint32_t blocked_offset[5]; // float offsets, [0] = 4^2 ... [4] = 1024^2
};

struct SceneRec {
    const RMat* rmats;          // resolved materials (index = material_id)
    uint32_t n_materials, n_instances;
    const pt_light* lights;
    const InstanceRec* instances;
    const Bvh4Node* nodes;
    const TriPacket* tris;
    const ShadePacket* shade;   // [triangle] parallel to tris
    int32_t root;              // wide node index (0), or ~0 when the scene is a single triangle
    uint32_t num_tris;
    const float* sheen_e;       // 16x16
    const float* srgb_lut;      // 256
    const float2* tangent_lut;  // 1024 x (sin, cos) of the packed tangent angle (pt_shading.h tangent_sincos_compute)
    EnvRec env;
    int32_t has_env;
    // Deep traversal stack, only for trees whose build-time bound exceeds the 64 entries a lane holds in LDS + scratch (long chains of
    // coincident centroids): entry k >= 64 of lane g lives at deep_stack[(k - 64) * deep_lanes + g], g = blockIdx.x * 256 + threadIdx.x.
    // deep_entries == 0 (every ordinary scene): one scalar test per pop / slow push, nothing else.
    int32_t* deep_stack;
    uint32_t deep_entries, deep_lanes;
    uint32_t small_tables;            // 1: every instance row, material and light of the scene is in the shade stage's LDS copies (set as a CONSTANT by the kernel
                                      // copy compiled for such scenes, k_wf_shade<.., true>: the table lookups then lose their global-memory branch); 0 elsewhere
};

// SceneConstants (PathTracer.lib.hlsl:10-30) plus the tile shard of this rank.
// x / d for x < 2^31 by a multiplication: mul = ceil(2^(31 + L) / d), L = ceil(log2 d), q = mulhi(x, mul) >> (L - 1) (Granlund-Montgomery
// round-up: exact for every x below 2^31; checked against x / d in tests/test_host_abi.py).  The kernels divide slot numbers by two
// per-frame constants in four places per path vertex; the hardware has no integer divider (a udiv is ~35 vector instructions).
struct FastDiv {
    uint32_t mul, shift, d;
    static FastDiv make(uint32_t d) {                      // host
        FastDiv f; f.d = d; f.mul = 0; f.shift = 0;
        if (d > 1) { uint32_t L = 0; while (((uint64_t)1 << L) < d) L++; f.mul = (uint32_t)((((uint64_t)1 << (31 + L)) + d - 1) / d); f.shift = L - 1; }
        return f;
    }
};

struct FrameConstants {
    float clip_to_world[16];
    float camera_pos[3];
    int32_t num_of_lights;
    uint32_t res_x, res_y, seed;
    int32_t accumulated_frames;
    float environment_color[3];
    float environment_intensity;
    int32_t debug_output;
    uint32_t flags;
    float max_ray_length;
    int32_t min_bounces, max_bounces;
    float luminance_clamp, min_rr, max_rr;
    uint32_t tiles_x, tiles_y;       // 16x16 tiles
    uint32_t tile_rank, tile_rank_count;
    uint32_t my_tiles;               // tiles this rank renders
    // Sample batch: one launch carries `spp` samples of every pixel (slot = sample * pixel_slots + pixel slot), equivalent to
    // `spp` consecutive PathtraceScene calls: sample k uses seed + k * seed_step and blends with accumulated_frames + k.
    uint32_t spp, pixel_slots, seed_step;
    uint32_t cull_null_shadow;       // pt_set_null_shadow_culling: do not trace a shadow ray whose pending term is exactly zero
    uint32_t defer_rare;             // the shade stage sets hits on rare materials aside and shades them together (k_wf_shade; set by the host
                                     // when a FEW of the scene's materials have the feature: with none there is nothing to gain, with many nothing either)
    FastDiv div_pixel_slots, div_tiles_x;   // divisions by pixel_slots / tiles_x (slot_pixel, slot_sample)
};

struct Counters {
    unsigned long long rays_primary, rays_bounce, rays_shadow, nodes, tris, hits, taps, stack_overflow;
    unsigned long long nodes_shadow, tris_shadow;      // the occlusion stage's share (wavefront mode; the megakernel books everything above)
    unsigned long long deep_pushes;                    // stack entries written beyond the 64 on-chip ones (SceneRec::deep_stack)
};

}  // namespace pt
