// pt_host.h -- host-visible declarations shared by the translation units of libmipt.so.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "pt_types.h"

namespace pt {

// ---- accel.hip --------------------------------------------------------------------------------
struct AccelScratch {
    TriPacket* tris_unsorted = nullptr;
    uint64_t *keys_a = nullptr, *keys_b = nullptr;
    uint32_t *vals_a = nullptr, *vals_b = nullptr;
    int32_t* leaf_parent = nullptr;
    int32_t* node_parent = nullptr;
    void* seg = nullptr;             // min/max segment tree over the sorted triangles' boxes (2 * seg_leaves entries of 32 B)
    size_t seg_leaves = 0;           // power of two >= capacity
    float* block_bounds = nullptr;   // k_setup's per-block centroid bounds
    uint32_t* bounds = nullptr;      // 6 sortable-uint floats: min xyz, max xyz
    void* sort_temp = nullptr;
    size_t sort_temp_bytes = 0;
    BvhNode* nodes2 = nullptr;       // the binary LBVH (intermediate)
    uint32_t* kept = nullptr;        // collapse frontier (ping): binary nodes that become wide nodes
    uint32_t* widx = nullptr;        // ... and the wide-node index each was given
    uint32_t* collapse_counters = nullptr;   // greedy collapse: [0] wide nodes allocated, [1 + L] frontier size of level L
    WideRanges* wide_ranges = nullptr;       // per wide node: the sorted-triangle range under each child (kept for accel_refit)
    // PLOC builder (accel.hip 4b)
    int builder = 2;                         // PT_BUILDER_*: 0 radix tree (Karras LBVH); 1 PLOC clustering over the same Morton order; 2 PLOC + reinsertion passes (default)
    float reinsert_min_gain = 1e-3f;         // ... a move must gain this fraction of its parent's surface area
    int reinsert_passes = 8;                 // builder 2: passes of parallel reinsertion over the clustered tree (accel.hip 4c)
    void* reins_box = nullptr; int32_t *reins_parent = nullptr, *reins_top = nullptr; float* reins_gain = nullptr; uint32_t* reins_out = nullptr;
    unsigned long long* reins_lock = nullptr; uint8_t* reins_state = nullptr;
    void* ploc_c[2] = {nullptr, nullptr};    // cluster arrays (ping-pong)
    uint32_t *ploc_nn = nullptr, *ploc_valid = nullptr, *ploc_pos = nullptr, *ploc_count = nullptr, *ploc_perm = nullptr, *ploc_counters = nullptr;
    int32_t *ploc_left = nullptr, *ploc_right = nullptr;
    void* ploc_scan_temp = nullptr; size_t ploc_scan_bytes = 0;
    size_t capacity = 0;
    std::string why;                         // what a failed build ran into (the hipError_t alone says "unknown error")
    uint32_t fallbacks = 0;                  // builds in which the PLOC clustering gave up and the radix tree took over
    std::string fallback_why;                // ... and why, the last time
};
void accel_scratch_free(AccelScratch& s);
// Builds the 4-wide BVH (<= n_tris nodes), the sorted intersection packets and their shading packets (n_tris each).
// root_out: 0, or ~0 for a single triangle.  wide_nodes_out: the node count; stack_need_out: the most traversal-stack entries any
// ray can hold in this tree (max over nodes of the siblings pushed on the way down).  Synchronises the stream (a full build is
// the rare event).
hipError_t accel_build(AccelScratch& s, const BufferRec* d_buffers, const InstanceRec* d_instances, int n_inst, uint32_t n_tris, Bvh4Node* d_nodes,
                       TriPacket* d_tris, ShadePacket* d_shade, int32_t* root_out, uint32_t* wide_nodes_out, uint32_t* stack_need_out, hipStream_t stream);
// Refit after vertices or instance transforms changed (topology and triangle order kept): rewrites the intersection and shading
// packets of the instances marked in d_touched[instance], rebuilds the segment tree over the triangles' boxes and requantises
// every wide node from it.  Asynchronous.  Hits equal those of a full rebuild; only the tree's quality follows the old pose.
hipError_t accel_refit(AccelScratch& s, const InstanceRec* d_instances, const uint8_t* d_touched, uint32_t n_tris, uint32_t wide_nodes,
                       Bvh4Node* d_nodes, TriPacket* d_tris, ShadePacket* d_shade, hipStream_t stream);

// ---- pt_kernel.hip: 1024 x (sin, cos) of the packed tangent angle
hipError_t build_tangent_lut(float2* d_lut, hipStream_t stream);

// ---- envmap.hip -------------------------------------------------------------------------------
struct EnvDevice {
    uint16_t* cube = nullptr;          // all mips, RGBA16F
    size_t mip_offset[16] = {0};       // in halfs
    int mip_n[16] = {0};
    int mips = 0;
    float* importance = nullptr;       // sum pyramid
    uint32_t level_offset[12] = {0};
    int levels = 0;
    int imp_res = 1024;
    float* blocked = nullptr;          // 4x4-blocked copies of levels 4^2, 16^2, 64^2, 256^2, 1024^2 (EnvRec::blocked)
    uint32_t blocked_offset[5] = {0};
    float total = 0.f;                 // the pyramid's apex (sum of the whole map), read back once: a kernel argument instead of a load per sample
};
hipError_t env_build(EnvDevice& e, const float* d_equirect, int w, int h, hipStream_t stream);
void env_free(EnvDevice& e);

// ---- skin_tonemap.hip -------------------------------------------------------------------------
struct SkinArgs {
    uint32_t num_of_vertices, input_mesh_flags, output_mesh_flags;
    int32_t num_of_morph_targets;
    float morph_weight[4];
    const float* morph_position[4];
    const uint32_t* morph_tangent_space[4];
    const float* in_position;
    const uint32_t* in_tangent_space;
    const uint4* in_joint_weight;
    const pt_bone* bones;
    int32_t bone_count;
    float* out_position;
    uint32_t* out_tangent_space;
};
void launch_skin(const SkinArgs& a, bool use_mfma, hipStream_t stream);
void launch_tonemap(const float4* in, uint32_t w, uint32_t h, const pt_tonemap_config& cfg, float* out_rgb, uint32_t* out_rgba8, hipStream_t stream);

// ---- pt_wavefront.hip / pt_kernel.hip ---------------------------------------------------------
size_t traversal_grid_lanes(int stage_blocks);   // lanes of the widest traversal launch of a wavefront trace with that many stage workgroups
constexpr uint32_t kDeepStackMax = 1024;      // most stack entries a tree may ask for (64 on chip + a deep stack in memory); a clustered tree beyond it is rebuilt as a radix tree
int traversal_stack_capacity();          // entries a ray's traversal stack holds on chip (LDS part + scratch spill); deeper trees get SceneRec::deep_stack
// Optional per-stage timing of one wavefront launch (pt_enable_stage_timing): an event after every stage launch.
enum { STAGE_GENERATE = 0, STAGE_TRACE = 1, STAGE_SHADE = 2, STAGE_SHADOW = 3, STAGE_RESOLVE = 4, STAGE_COUNT = 5 };
struct StageTimers {
    std::vector<hipEvent_t> ev;          // ev[0] = start; ev[k + 1] = after the k-th launch
    std::vector<uint8_t> kind;           // STAGE_* of the k-th launch
    size_t used = 0;                     // launches recorded by the last pt_trace
};
void launch_debug_intersect(const SceneRec& sc, const float* d_rays, uint32_t n, uint32_t rf, int mode, float* d_out, hipStream_t stream);
void launch_megakernel(const SceneRec& sc, const FrameConstants& fc, float4* output, Counters* counters, bool count, hipStream_t stream);
size_t wavefront_workspace_bytes(const FrameConstants& fc, int stage_blocks);
// occ_cache: the context's occluder cache (res_x * res_y * 8 words, persistent across calls; nullptr = none), see WfBuffers::occ_cache
hipError_t launch_wavefront(const SceneRec& sc, const FrameConstants& fc, float4* output, Counters* counters, bool count, void* workspace,
                            int stage_blocks, StageTimers* timers, hipStream_t stream, uint32_t* occ_cache);

// ---- sort_scan.hip: the build's two data-parallel primitives, hand-written (stable LSD radix sort of (u64, u32) pairs over 63 key bits; u32 exclusive scan)
size_t radix_sort_temp_bytes(size_t n);
hipError_t radix_sort_pairs_u64_u32(void* temp, uint64_t* keys_in, uint64_t* keys_out, uint32_t* vals_in, uint32_t* vals_out, size_t n, hipStream_t stream);   // keys_in / vals_in are scratch
size_t exclusive_scan_temp_bytes(size_t n);
hipError_t exclusive_scan_u32(void* temp, const uint32_t* in, uint32_t* out, size_t n, hipStream_t stream);

// ---- exchange.hip: the per-frame tile exchange of the sharded renderer (RCCL bound at run time) ---------------------------
struct ExchangeState;
uint32_t tiles_of_rank(uint32_t w, uint32_t h, uint32_t rank, uint32_t world);
hipError_t tiles_pack(const void* image, uint32_t w, uint32_t h, uint32_t rank, uint32_t world, void* packed, hipStream_t stream);
hipError_t tiles_unpack(const void* packed, uint32_t w, uint32_t h, uint32_t rank, uint32_t world, void* image, hipStream_t stream);
int exchange_unique_id(void* out128, std::string& err);
int exchange_probe(std::string& err);
int exchange_create(ExchangeState** out, int rank, int world, const void* id128, std::string& err);
int exchange_create_loopback(ExchangeState** out, int rank, int world, uint64_t group, int device, std::string& err);
const char* exchange_transport_name(const ExchangeState* x);
int exchange_frame(ExchangeState* x, const void* local, void* frame, uint32_t w, uint32_t h, int mode, int dst, hipStream_t stream, std::string& err);
void exchange_free(ExchangeState* x);

}  // namespace pt
