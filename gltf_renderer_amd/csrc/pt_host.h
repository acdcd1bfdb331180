// pt_host.h -- host-visible declarations shared by the translation units of libmipt.so.
#pragma once
#include <hip/hip_runtime.h>

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
    size_t capacity = 0;
};
void accel_scratch_free(AccelScratch& s);
// Builds the 4-wide BVH (<= n_tris nodes), the sorted intersection packets and their shading packets (n_tris each).
// root_out: 0, or ~0 for a single triangle.  wide_nodes_out: the node count, or kWideNodesOnDevice when the build did not wait
// for it (small scenes): it is then s.collapse_counters[0] once the stream has caught up.
constexpr uint32_t kWideNodesOnDevice = 0xffffffffu;
hipError_t accel_build(AccelScratch& s, const BufferRec* d_buffers, const InstanceRec* d_instances, int n_inst, uint32_t n_tris, Bvh4Node* d_nodes,
                       TriPacket* d_tris, ShadePacket* d_shade, int32_t* root_out, uint32_t* wide_nodes_out, hipStream_t stream);

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

// ---- pt_kernel.hip ----------------------------------------------------------------------------
void launch_megakernel(const SceneRec& sc, const FrameConstants& fc, float4* output, Counters* counters, bool count, hipStream_t stream);
size_t wavefront_workspace_bytes(uint32_t slots, int stage_blocks);
hipError_t launch_wavefront(const SceneRec& sc, const FrameConstants& fc, float4* output, Counters* counters, bool count, void* workspace,
                            int stage_blocks, hipStream_t stream);

}  // namespace pt
