// mipt_api.hip -- host side of libmipt.so: the C-ABI of include/mipt.h over C++ mirrors of the
// reference's hot-path classes.
//
//   class Pathtracer   <- Source/Pathtracer.{h,cpp}: Init / PathtraceScene / Shutdown, the cross-frame
//                         state (accumulated_frames, previous_world_to_clip), BuildAllBlas/UpdateAllBlas/BuildTlas
//                         folded into one on-device LBVH build (accel.hip)
//   class GpuSkin      <- Source/GpuSkin.{h,cpp}: Create / Run
//   class EnvironmentMap <- Source/EnvironmentMap.{h,cpp}: CreateEnvironmentMap (cube + importance only)
//   ResourceTable      <- the bindless descriptor heap (DescriptorAllocator.h), as raw device pointers
// There is no CPU fallback anywhere in this file: every compute entry point launches HIP kernels.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "pt_types.h"

#include "pt_host.h"

using namespace pt;

struct pt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string error;

    // ---- ResourceTable ("descriptor heap")
    std::vector<BufferRec> buffers;
    std::vector<TextureRec> textures;
    std::vector<SamplerRec> samplers;
    BufferRec* d_buffers = nullptr; size_t d_buffers_cap = 0; bool buffers_dirty = true;   // device copy: BVH build only
    uint32_t* d_white = nullptr;                                   // 1x1 white texel behind every unbound material slot

    // ---- per-frame arrays (Renderer::GatherMaterials / GatherLights)
    RMat* d_rmats = nullptr; int n_materials = 0; size_t rmats_cap = 0;    // resolved on the host in pt_scene_set_materials
    pt_light* d_lights = nullptr; int n_lights = 0; size_t lights_cap = 0;

    // ---- instance table + acceleration structure
    std::vector<InstanceRec> instances;
    InstanceRec* d_instances = nullptr; size_t instances_cap = 0;
    uint32_t n_tris = 0;
    Bvh4Node* d_nodes = nullptr; TriPacket* d_tris = nullptr; ShadePacket* d_shade = nullptr; size_t accel_cap = 0;
    uint32_t wide_nodes = 0;
    int32_t root = 0;
    AccelScratch scratch;
    bool accel_dirty = true;

    std::vector<EnvDevice*> envs;
    float* d_sheen = nullptr;
    float* d_srgb = nullptr;
    float2* d_tangent_lut = nullptr;
    Counters* d_counters = nullptr;
    void* d_bones = nullptr; size_t bones_cap = 0;
    void* d_workspace = nullptr; size_t workspace_cap = 0;     // wavefront ray / hit / path-state arrays
    int kernel_mode = PT_MODE_WAVEFRONT;
    int stage_blocks = 0;         // workgroups per stage launch; 0 = by the size of the launch (stage_blocks_for)
    hipEvent_t ev_trace[2] = {nullptr, nullptr}, ev_accel[2] = {nullptr, nullptr}, ev_skin[2] = {nullptr, nullptr};
    bool have_trace = false, have_accel = false, have_skin = false;
    int bounce_limit = PT_REFERENCE_MAX_BOUNCES;
    int samples_per_trace = 1;
    bool cull_null_shadow = false;
    bool counters_enabled = false;

    // ---- Pathtracer cross-frame state (Source/Pathtracer.h:152-153)
    float previous_world_to_clip[16] = {0};
    int accumulated_frames = 0;

    int fail(int code, const std::string& msg) { error = msg; return code; }
};

#define HIPOK(call)                                                                                          \
    do {                                                                                                     \
        hipError_t _e = (call);                                                                              \
        if (_e != hipSuccess) return ctx->fail(PT_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(_e)); \
    } while (0)

namespace {

template <typename T>
hipError_t upload_table(T*& d, size_t& cap, const std::vector<T>& h, hipStream_t s) {
    size_t n = h.size() ? h.size() : 1;
    if (n > cap) {
        hipFree(d);
        d = nullptr;
        size_t nc = n + n / 2 + 8;
        hipError_t e = hipMalloc((void**)&d, nc * sizeof(T));
        if (e) return e;
        cap = nc;
    }
    if (h.empty()) return hipSuccess;
    hipError_t e = hipMemcpyAsync(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s);
    if (e) return e;
    return hipStreamSynchronize(s);      // h may be a temporary of the caller's frame (transient heap semantics)
}

// glm closed forms.  Inverses in fp64, rounded once.
void mat4_mul(const float* a, const float* b, float* out) {
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++) {
            float s = 0;
            for (int k = 0; k < 4; k++) s += a[k * 4 + r] * b[c * 4 + k];
            out[c * 4 + r] = s;
        }
}
bool mat4_inverse(const float* mf, float* out) {
    double m[16], inv[16];
    for (int i = 0; i < 16; i++) m[i] = mf[i];
    double s0 = m[0] * m[5] - m[4] * m[1], s1 = m[0] * m[9] - m[8] * m[1], s2 = m[0] * m[13] - m[12] * m[1];
    double s3 = m[4] * m[9] - m[8] * m[5], s4 = m[4] * m[13] - m[12] * m[5], s5 = m[8] * m[13] - m[12] * m[9];
    double c5 = m[10] * m[15] - m[14] * m[11], c4 = m[6] * m[15] - m[14] * m[7], c3 = m[6] * m[11] - m[10] * m[7];
    double c2 = m[2] * m[15] - m[14] * m[3], c1 = m[2] * m[11] - m[10] * m[3], c0 = m[2] * m[7] - m[6] * m[3];
    double det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
    if (det == 0) return false;
    double id = 1.0 / det;
    inv[0] = (m[5] * c5 - m[9] * c4 + m[13] * c3) * id;
    inv[4] = (-m[4] * c5 + m[8] * c4 - m[12] * c3) * id;
    inv[8] = (m[7] * s5 - m[11] * s4 + m[15] * s3) * id;
    inv[12] = (-m[6] * s5 + m[10] * s4 - m[14] * s3) * id;
    inv[1] = (-m[1] * c5 + m[9] * c2 - m[13] * c1) * id;
    inv[5] = (m[0] * c5 - m[8] * c2 + m[12] * c1) * id;
    inv[9] = (-m[3] * s5 + m[11] * s2 - m[15] * s1) * id;
    inv[13] = (m[2] * s5 - m[10] * s2 + m[14] * s1) * id;
    inv[2] = (m[1] * c4 - m[5] * c2 + m[13] * c0) * id;
    inv[6] = (-m[0] * c4 + m[4] * c2 - m[12] * c0) * id;
    inv[10] = (m[3] * s4 - m[7] * s2 + m[15] * s0) * id;
    inv[14] = (-m[2] * s4 + m[6] * s2 - m[14] * s0) * id;
    inv[3] = (-m[1] * c3 + m[5] * c1 - m[9] * c0) * id;
    inv[7] = (m[0] * c3 - m[4] * c1 + m[8] * c0) * id;
    inv[11] = (-m[3] * s3 + m[7] * s1 - m[11] * s0) * id;
    inv[15] = (m[2] * s3 - m[6] * s1 + m[10] * s0) * id;
    for (int i = 0; i < 16; i++) out[i] = (float)inv[i];
    return true;
}

size_t format_stride(int f) {
    switch (f) {
        case PT_FORMAT_R16_UINT: return 2;
        case PT_FORMAT_R32_UINT: return 4;
        case PT_FORMAT_R32G32B32_FLOAT: return 12;
        case PT_FORMAT_R10G10B10A2_UNORM: return 4;
        case PT_FORMAT_R32G32_FLOAT: return 8;
        case PT_FORMAT_R16G16B16A16_UNORM: return 8;
        case PT_FORMAT_JOINT_WEIGHT: return 16;
        default: return 0;
    }
}

// Workgroups per stage launch.  256 CUs hold 6 resident 256-thread workgroups of the trace stages (LDS- and VGPR-limited): 1536
// for a launch that has the rays to feed them.  A small launch (a 1/8 tile shard of one sample) is better off with fewer
// persistent workgroups -- each stages its LDS tables and polls the shard queues whether it gets rays or not.  Measured
// (tools/stage_blocks_probe.py, Mrays/s at 512 / 768 / 1536 workgroups): 259 k slots 1121 / 1099 / 1065, 518 k slots
// 1632 / 1685 / 1551, 2.07 M slots 2500 / 2697 / 3149.
int stage_blocks_for(size_t slots) { return slots >= 1200000 ? 1536 : (slots >= 400000 ? 768 : 512); }

}  // namespace

// =================================================================================================
// class Pathtracer (Source/Pathtracer.h:16-157)
namespace pt {
class Pathtracer {
public:
    // Pathtracer::BuildAllBlas + UpdateAllBlas + BuildTlas (Source/Pathtracer.cpp:138-257)
    static int BuildAccel(pt_ctx* ctx) {
        if (ctx->buffers_dirty) { HIPOK(upload_table(ctx->d_buffers, ctx->d_buffers_cap, ctx->buffers, ctx->stream)); ctx->buffers_dirty = false; }
        HIPOK(upload_table(ctx->d_instances, ctx->instances_cap, ctx->instances, ctx->stream));
        size_t need = ctx->n_tris ? ctx->n_tris : 1;
        if (need > ctx->accel_cap) {
            hipFree(ctx->d_nodes); hipFree(ctx->d_tris); hipFree(ctx->d_shade);
            ctx->d_nodes = nullptr; ctx->d_tris = nullptr; ctx->d_shade = nullptr;
            size_t cap = need + need / 8 + 64;
            HIPOK(hipMalloc((void**)&ctx->d_nodes, cap * sizeof(Bvh4Node)));
            HIPOK(hipMalloc((void**)&ctx->d_tris, cap * sizeof(TriPacket)));
            HIPOK(hipMalloc((void**)&ctx->d_shade, cap * sizeof(ShadePacket)));
            ctx->accel_cap = cap;
        }
        HIPOK(hipEventRecord(ctx->ev_accel[0], ctx->stream));
        HIPOK(accel_build(ctx->scratch, ctx->d_buffers, ctx->d_instances, (int)ctx->instances.size(), ctx->n_tris, ctx->d_nodes, ctx->d_tris, ctx->d_shade,
                          &ctx->root, &ctx->wide_nodes, ctx->stream));
        HIPOK(hipEventRecord(ctx->ev_accel[1], ctx->stream));
        ctx->have_accel = true;
        ctx->accel_dirty = false;
        return PT_OK;
    }

    // Pathtracer::PathtraceScene (Source/Pathtracer.cpp:259-367)
    static int PathtraceScene(pt_ctx* ctx, const pt_settings* settings, const pt_execute_params* ep) {
        float world_to_clip[16], clip_to_world[16], view_to_world[16];
        mat4_mul(ep->view_to_clip, ep->world_to_view, world_to_clip);                   // :262
        if (!mat4_inverse(ep->world_to_view, view_to_world) || !mat4_inverse(world_to_clip, clip_to_world))
            return ctx->fail(PT_ERR_INVALID_ARGUMENT, "singular camera matrix");
        bool reset = memcmp(world_to_clip, ctx->previous_world_to_clip, 64) != 0 || settings->reset;   // :267-271
        if (reset) ctx->accumulated_frames = 0;
        if (ctx->accumulated_frames < settings->max_accumulated_frames) {               // :273
            if (ep->light_count > ctx->n_lights) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "light_count exceeds uploaded lights");
            if (ep->environment_map >= (int)ctx->envs.size()) return ctx->fail(PT_ERR_BAD_HANDLE, "bad environment map handle");
            if (!ep->output) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "output is null");
            if (ctx->accel_dirty) { int r = BuildAccel(ctx); if (r) return r; }

            SceneRec sc;
            memset(&sc, 0, sizeof(sc));
            sc.rmats = ctx->d_rmats; sc.lights = ctx->d_lights; sc.instances = ctx->d_instances;
            sc.n_materials = (uint32_t)ctx->n_materials; sc.n_instances = (uint32_t)ctx->instances.size();
            sc.nodes = ctx->d_nodes; sc.tris = ctx->d_tris; sc.shade = ctx->d_shade; sc.root = ctx->root; sc.num_tris = ctx->n_tris;
            sc.sheen_e = ctx->d_sheen; sc.srgb_lut = ctx->d_srgb; sc.tangent_lut = ctx->d_tangent_lut;
            sc.has_env = 0;
            if (ep->environment_map >= 0) {
                const EnvDevice& ed = *ctx->envs[ep->environment_map];
                sc.env.cube = ed.cube; sc.env.cube_n = ed.mip_n[0]; sc.env.importance = ed.importance;
                for (int i = 0; i < 12; i++) sc.env.level_offset[i] = ed.level_offset[i];
                sc.env.imp_res = ed.imp_res; sc.env.imp_levels = ed.levels; sc.env.imp_total = ed.total;
                sc.env.blocked = ed.blocked;
                for (int i = 0; i < 5; i++) sc.env.blocked_offset[i] = ed.blocked_offset[i];
                sc.has_env = 1;
            }

            FrameConstants fc;                                                          // :287-331
            memset(&fc, 0, sizeof(fc));
            memcpy(fc.clip_to_world, clip_to_world, 64);
            fc.camera_pos[0] = view_to_world[12]; fc.camera_pos[1] = view_to_world[13]; fc.camera_pos[2] = view_to_world[14];
            fc.num_of_lights = ep->light_count;
            fc.res_x = ep->width; fc.res_y = ep->height;
            fc.seed = settings->use_frame_as_seed ? (uint32_t)ep->frame : settings->seed;   // :316
            fc.accumulated_frames = ctx->accumulated_frames;
            memcpy(fc.environment_color, settings->environment_color, 12);
            fc.environment_intensity = settings->environment_intensity;
            fc.debug_output = settings->debug_output;
            fc.flags = settings->flags;
            fc.max_ray_length = 1000;                                                   // :322 (the setting is ignored)
            auto clampi = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
            fc.min_bounces = clampi(settings->min_bounces, 0, ctx->bounce_limit);        // :323-324
            fc.max_bounces = clampi(settings->max_bounces, 0, ctx->bounce_limit);
            fc.luminance_clamp = settings->luminance_clamp;
            fc.min_rr = settings->min_russian_roulette_continue_prob;
            fc.max_rr = settings->max_russian_roulette_continue_prob;
            fc.tiles_x = (ep->width + PT_TILE - 1) / PT_TILE;
            fc.tiles_y = (ep->height + PT_TILE - 1) / PT_TILE;
            fc.tile_rank_count = ep->tile_rank_count ? ep->tile_rank_count : 1;
            fc.tile_rank = ep->tile_rank;
            if (fc.tile_rank >= fc.tile_rank_count) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "tile_rank >= tile_rank_count");
            uint32_t ntiles = fc.tiles_x * fc.tiles_y;
            fc.my_tiles = ntiles > fc.tile_rank ? (ntiles - fc.tile_rank + fc.tile_rank_count - 1) / fc.tile_rank_count : 0;

            // Sample batch (pt_set_samples_per_trace): this call stands for `batch` consecutive PathtraceScene calls of frames
            // frame .. frame + batch - 1 with an unchanged camera.  Only accumulation makes more than the last one observable,
            // and the batch may not run past max_accumulated_frames (the calls beyond it would have been no-ops, :273).
            int batch = 1;
            if ((settings->flags & PT_FLAG_ACCUMULATE) && settings->debug_output == PT_DEBUG_OUTPUT_NONE) {
                batch = ctx->samples_per_trace;
                const long long room = (long long)settings->max_accumulated_frames - ctx->accumulated_frames;
                if ((long long)batch > room) batch = (int)room;
            }
            fc.cull_null_shadow = ctx->cull_null_shadow ? 1u : 0u;
            fc.spp = 1; fc.pixel_slots = fc.my_tiles * 256u;
            fc.seed_step = settings->use_frame_as_seed ? 1u : 0u;

            HIPOK(hipEventRecord(ctx->ev_trace[0], ctx->stream));
            if (ctx->kernel_mode == PT_MODE_MEGAKERNEL) {
                for (int k = 0; k < batch; k++) {                                        // the megakernel has no batch form: one launch per sample
                    FrameConstants fk = fc;
                    fk.seed = fc.seed + (uint32_t)k * fc.seed_step;
                    fk.accumulated_frames = fc.accumulated_frames + k;
                    launch_megakernel(sc, fk, (float4*)ep->output, ctx->d_counters, ctx->counters_enabled, ctx->stream);   // :344-353
                }
            } else {
                fc.spp = (uint32_t)batch;
                if ((unsigned long long)fc.pixel_slots * fc.spp > 0x7fffffffull) return ctx->fail(PT_ERR_CAPACITY, "sample batch too large for this resolution");
                const int stage_blocks = ctx->stage_blocks > 0 ? ctx->stage_blocks : stage_blocks_for((size_t)fc.pixel_slots * fc.spp);
                size_t need = wavefront_workspace_bytes(fc.pixel_slots * fc.spp, stage_blocks);
                if (need > ctx->workspace_cap) {
                    HIPOK(hipStreamSynchronize(ctx->stream));
                    hipFree(ctx->d_workspace); ctx->d_workspace = nullptr; ctx->workspace_cap = 0;
                    HIPOK(hipMalloc(&ctx->d_workspace, need));
                    ctx->workspace_cap = need;
                }
                HIPOK(launch_wavefront(sc, fc, (float4*)ep->output, ctx->d_counters, ctx->counters_enabled, ctx->d_workspace, stage_blocks, ctx->stream));
            }
            HIPOK(hipGetLastError());
            HIPOK(hipEventRecord(ctx->ev_trace[1], ctx->stream));
            ctx->have_trace = true;
            if (settings->flags & PT_FLAG_ACCUMULATE) ctx->accumulated_frames += batch;   // :355-359
            else ctx->accumulated_frames = 0;
        }
        memcpy(ctx->previous_world_to_clip, world_to_clip, 64);                          // :366
        return PT_OK;
    }
};

// class GpuSkin (Source/GpuSkin.h:9-37)
class GpuSkin {
public:
    static int Run(pt_ctx* ctx, const pt_skin_params* p, const pt_bone* bones, int bone_count) {      // GpuSkin.cpp:57-118
        auto buf = [&](int h, int fmt, size_t count, const void** out) -> bool {
            if (h < 0 || h >= (int)ctx->buffers.size()) return false;
            const BufferRec& b = ctx->buffers[h];
            if ((int)b.format != fmt || b.bytes < count * format_stride(fmt)) return false;
            *out = b.ptr;
            return true;
        };
        SkinArgs a;
        memset(&a, 0, sizeof(a));
        a.num_of_vertices = p->num_of_vertices;
        a.input_mesh_flags = p->input_mesh_flags;
        a.output_mesh_flags = p->output_mesh_flags;
        a.num_of_morph_targets = p->num_of_morph_targets < PT_MAX_SIMULTANEOUS_MORPH_TARGETS ? p->num_of_morph_targets : PT_MAX_SIMULTANEOUS_MORPH_TARGETS;
        if (a.num_of_morph_targets < 0) a.num_of_morph_targets = 0;
        // If no bones are supplied, `input_mesh_flags &= !FLAG_JOINT_WEIGHT` clears ALL input flags (quirk q19, GpuSkin.cpp:94)
        if (!bones || bone_count <= 0) a.input_mesh_flags &= (uint32_t)!PT_MESH_FLAG_JOINT_WEIGHT;
        const void* ptr = nullptr;
        if (!buf(p->input_position, PT_FORMAT_R32G32B32_FLOAT, p->num_of_vertices, &ptr)) return ctx->fail(PT_ERR_BAD_HANDLE, "skin: input_position");
        a.in_position = (const float*)ptr;
        if (a.input_mesh_flags & PT_MESH_FLAG_TANGENT_SPACE) {
            if (!buf(p->input_tangent_space, PT_FORMAT_R10G10B10A2_UNORM, p->num_of_vertices, &ptr)) return ctx->fail(PT_ERR_BAD_HANDLE, "skin: input_tangent_space");
            a.in_tangent_space = (const uint32_t*)ptr;
        }
        if (a.input_mesh_flags & PT_MESH_FLAG_JOINT_WEIGHT) {
            if (!buf(p->input_joint_weight, PT_FORMAT_JOINT_WEIGHT, p->num_of_vertices, &ptr)) return ctx->fail(PT_ERR_BAD_HANDLE, "skin: input_joint_weight");
            a.in_joint_weight = (const uint4*)ptr;
        }
        if (a.output_mesh_flags & PT_DYNAMIC_MESH_FLAG_POSITION) {
            if (!buf(p->output_position, PT_FORMAT_R32G32B32_FLOAT, p->num_of_vertices, &ptr)) return ctx->fail(PT_ERR_BAD_HANDLE, "skin: output_position");
            a.out_position = (float*)ptr;
        }
        if (a.output_mesh_flags & PT_DYNAMIC_MESH_FLAG_TANGENT_SPACE) {
            if (!buf(p->output_tangent_space, PT_FORMAT_R10G10B10A2_UNORM, p->num_of_vertices, &ptr)) return ctx->fail(PT_ERR_BAD_HANDLE, "skin: output_tangent_space");
            a.out_tangent_space = (uint32_t*)ptr;
        }
        for (int i = 0; i < a.num_of_morph_targets; i++) {
            a.morph_weight[i] = p->morph_weights[i];
            if (p->morph_position[i] != -1) {
                if (!buf(p->morph_position[i], PT_FORMAT_R32G32B32_FLOAT, p->num_of_vertices, &ptr)) return ctx->fail(PT_ERR_BAD_HANDLE, "skin: morph_position");
                a.morph_position[i] = (const float*)ptr;
            }
            if (p->morph_tangent_space[i] != -1) {
                if (!buf(p->morph_tangent_space[i], PT_FORMAT_R10G10B10A2_UNORM, p->num_of_vertices, &ptr)) return ctx->fail(PT_ERR_BAD_HANDLE, "skin: morph_tangent_space");
                a.morph_tangent_space[i] = (const uint32_t*)ptr;
            }
        }
        if (a.input_mesh_flags & PT_MESH_FLAG_JOINT_WEIGHT) {
            size_t bytes = (size_t)bone_count * sizeof(pt_bone);
            if (bytes > ctx->bones_cap) {
                hipFree(ctx->d_bones); ctx->d_bones = nullptr; ctx->bones_cap = 0;
                HIPOK(hipMalloc(&ctx->d_bones, bytes * 2));
                ctx->bones_cap = bytes * 2;
            }
            HIPOK(hipMemcpyAsync(ctx->d_bones, bones, bytes, hipMemcpyHostToDevice, ctx->stream));
            HIPOK(hipStreamSynchronize(ctx->stream));        // bones live in the caller's transient heap (Renderer.cpp:411)
            a.bones = (const pt_bone*)ctx->d_bones;
            a.bone_count = bone_count;
        }
        HIPOK(hipEventRecord(ctx->ev_skin[0], ctx->stream));
        launch_skin(a, p->use_mfma != 0, ctx->stream);
        HIPOK(hipGetLastError());
        HIPOK(hipEventRecord(ctx->ev_skin[1], ctx->stream));
        ctx->have_skin = true;
        ctx->accel_dirty = true;                              // UpdateAllBlas runs every frame for dynamic meshes (Pathtracer.cpp:168-183)
        return PT_OK;
    }
};
}  // namespace pt

// =================================================================================================
// C-ABI
extern "C" {

int pt_abi_version(void) { return MIPT_ABI_VERSION; }

int pt_create(int device, void* hip_stream, const float* sheen_e_16x16, pt_ctx** out) {
    if (!out || !sheen_e_16x16) return PT_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return PT_ERR_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return PT_ERR_DEVICE;
    pt_ctx* ctx = new pt_ctx();
    ctx->device = device;
    ctx->stream = (hipStream_t)hip_stream;
    float srgb[256];
    for (int i = 0; i < 256; i++) {
        double c = i / 255.0;
        srgb[i] = (float)(c <= 0.04045 ? c / 12.92 : pow((c + 0.055) / 1.055, 2.4));
    }
    bool ok = hipMalloc((void**)&ctx->d_sheen, 256 * 4) == hipSuccess && hipMalloc((void**)&ctx->d_srgb, 256 * 4) == hipSuccess &&
              hipMalloc((void**)&ctx->d_counters, sizeof(Counters)) == hipSuccess && hipMalloc((void**)&ctx->d_white, 16) == hipSuccess &&
              hipMemset(ctx->d_white, 0xff, 16) == hipSuccess &&
              hipMemcpy(ctx->d_sheen, sheen_e_16x16, 256 * 4, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(ctx->d_srgb, srgb, 256 * 4, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemset(ctx->d_counters, 0, sizeof(Counters)) == hipSuccess &&
              hipMalloc((void**)&ctx->d_tangent_lut, 1024 * sizeof(float2)) == hipSuccess &&
              build_tangent_lut(ctx->d_tangent_lut, ctx->stream) == hipSuccess;
    for (int i = 0; i < 2 && ok; i++)
        ok = hipEventCreate(&ctx->ev_trace[i]) == hipSuccess && hipEventCreate(&ctx->ev_accel[i]) == hipSuccess && hipEventCreate(&ctx->ev_skin[i]) == hipSuccess;
    if (!ok) { pt_destroy(ctx); return PT_ERR_DEVICE; }
    ctx->samplers.push_back({PT_ADDRESS_WRAP, PT_ADDRESS_WRAP, PT_FILTER_LINEAR, PT_FILTER_LINEAR});   // sampler 0 (GpuResources.cpp:47-59)
    *out = ctx;
    return PT_OK;
}

void pt_destroy(pt_ctx* ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    for (auto& b : ctx->buffers) hipFree((void*)b.ptr);
    for (auto& t : ctx->textures) hipFree((void*)t.texels);
    for (auto* e : ctx->envs) { env_free(*e); delete e; }
    hipFree(ctx->d_buffers); hipFree(ctx->d_white); hipFree(ctx->d_rmats); hipFree(ctx->d_lights);
    hipFree(ctx->d_instances); hipFree(ctx->d_nodes); hipFree(ctx->d_tris); hipFree(ctx->d_shade); hipFree(ctx->d_sheen); hipFree(ctx->d_srgb); hipFree(ctx->d_tangent_lut); hipFree(ctx->d_counters);
    accel_scratch_free(ctx->scratch);
    hipFree(ctx->d_bones);
    hipFree(ctx->d_workspace);
    for (int i = 0; i < 2; i++) {
        if (ctx->ev_trace[i]) hipEventDestroy(ctx->ev_trace[i]);
        if (ctx->ev_accel[i]) hipEventDestroy(ctx->ev_accel[i]);
        if (ctx->ev_skin[i]) hipEventDestroy(ctx->ev_skin[i]);
    }
    delete ctx;
}

const char* pt_last_error(const pt_ctx* ctx) { return ctx ? ctx->error.c_str() : "null context"; }

int pt_buffer_create(pt_ctx* ctx, const void* host, size_t bytes, int format, int* handle_out) {
    if (!ctx || !handle_out) return PT_ERR_INVALID_ARGUMENT;
    if (format_stride(format) == 0 || bytes == 0 || bytes > 0xffffffffull) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "pt_buffer_create: bad format or size");
    void* d = nullptr;
    HIPOK(hipMalloc(&d, bytes + 16));
    if (host) HIPOK(hipMemcpy(d, host, bytes, hipMemcpyHostToDevice));
    else HIPOK(hipMemset(d, 0, bytes));
    ctx->buffers.push_back({d, (uint32_t)format, (uint32_t)bytes});
    ctx->buffers_dirty = true;
    *handle_out = (int)ctx->buffers.size() - 1;
    return PT_OK;
}

int pt_buffer_update(pt_ctx* ctx, int handle, const void* host, size_t bytes) {
    if (!ctx || !host) return PT_ERR_INVALID_ARGUMENT;
    if (handle < 0 || handle >= (int)ctx->buffers.size() || bytes > ctx->buffers[handle].bytes) return ctx->fail(PT_ERR_BAD_HANDLE, "pt_buffer_update");
    HIPOK(hipMemcpyAsync((void*)ctx->buffers[handle].ptr, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPOK(hipStreamSynchronize(ctx->stream));
    ctx->accel_dirty = true;
    return PT_OK;
}

int pt_buffer_read(pt_ctx* ctx, int handle, void* host, size_t bytes) {
    if (!ctx || !host) return PT_ERR_INVALID_ARGUMENT;
    if (handle < 0 || handle >= (int)ctx->buffers.size() || bytes > ctx->buffers[handle].bytes) return ctx->fail(PT_ERR_BAD_HANDLE, "pt_buffer_read");
    HIPOK(hipStreamSynchronize(ctx->stream));
    HIPOK(hipMemcpy(host, ctx->buffers[handle].ptr, bytes, hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_texture_create(pt_ctx* ctx, const uint8_t* rgba8, int width, int height, int srgb, int* handle_out) {
    if (!ctx || !rgba8 || !handle_out || width <= 0 || height <= 0) return PT_ERR_INVALID_ARGUMENT;
    void* d = nullptr;
    size_t bytes = (size_t)width * height * 4;
    HIPOK(hipMalloc(&d, bytes));
    HIPOK(hipMemcpy(d, rgba8, bytes, hipMemcpyHostToDevice));
    ctx->textures.push_back({(const uint32_t*)d, width, height, srgb ? 1u : 0u, 0u});
    *handle_out = (int)ctx->textures.size() - 1;
    return PT_OK;
}

int pt_sampler_create(pt_ctx* ctx, const pt_sampler_desc* d, int* handle_out) {
    if (!ctx || !d || !handle_out) return PT_ERR_INVALID_ARGUMENT;
    if (d->address_u < 0 || d->address_u > 2 || d->address_v < 0 || d->address_v > 2) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "pt_sampler_create: address mode");
    ctx->samplers.push_back({d->address_u, d->address_v, d->min_filter, d->mag_filter});
    *handle_out = (int)ctx->samplers.size() - 1;
    return PT_OK;
}

int pt_scene_set_materials(pt_ctx* ctx, const pt_material* m, int count) {
    if (!ctx || (count > 0 && !m) || count < 0) return PT_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < count; i++) {
        const pt_texture_sample* slots[15] = {&m[i].normal, &m[i].albedo, &m[i].metallic_roughness, &m[i].occlusion, &m[i].emissive, &m[i].specular,
                                              &m[i].specular_color, &m[i].clearcoat, &m[i].clearcoat_roughness, &m[i].clearcoat_normal, &m[i].anisotropy,
                                              &m[i].sheen_color, &m[i].sheen_roughness, &m[i].transmission, &m[i].thickness};
        for (auto* s : slots) {
            if (s->descriptor < -1 || s->descriptor >= (int)ctx->textures.size()) return ctx->fail(PT_ERR_BAD_HANDLE, "material texture descriptor out of range");
            if (s->sampler < 0 || s->sampler >= (int)ctx->samplers.size()) return ctx->fail(PT_ERR_BAD_HANDLE, "material sampler out of range");
        }
    }
    // Resolve every material into the kernel-side record (pt_types.h RMat): descriptor/sampler indices become pointers and
    // packed flags, the UV transform T*(R*S) (Material.hlsli:68-88) is multiplied out once in fp32.
    std::vector<RMat> rm((size_t)count);
    for (int i = 0; i < count; i++) {
        const pt_material& s = m[i];
        RMat& r = rm[i];
        memset(&r, 0, sizeof(r));
        r.flags = s.flags; r.alpha_mode = s.alpha_mode; r.metalness_factor = s.metalness_factor; r.roughness_factor = s.roughness_factor;
        memcpy(r.base_color_factor, s.base_color_factor, 16);
        memcpy(r.emissive_factor, s.emissive_factor, 12); r.alpha_cutoff = s.alpha_cutoff;
        r.ior = s.ior; r.normal_scale = s.normal_scale; r.specular_factor = s.specular_factor; r.clearcoat_normal_scale = s.clearcoat_normal_scale;
        memcpy(r.specular_color_factor, s.specular_color_factor, 12); r.clearcoat_factor = s.clearcoat_factor;
        r.clearcoat_roughness_factor = s.clearcoat_roughness_factor; r.anisotropy_strength = s.anisotropy_strength;
        r.anisotropy_cos = cosf(s.anisotropy_rotation); r.anisotropy_sin = sinf(s.anisotropy_rotation);
        memcpy(r.sheen_color_factor, s.sheen_color_factor, 12); r.sheen_roughness_factor = s.sheen_roughness_factor;
        r.transmission_factor = s.transmission_factor;
        const pt_texture_sample* slots[SLOT_COUNT] = {&s.normal, &s.albedo, &s.metallic_roughness, &s.occlusion, &s.emissive, &s.specular,
                                                      &s.specular_color, &s.clearcoat, &s.clearcoat_roughness, &s.clearcoat_normal, &s.anisotropy,
                                                      &s.sheen_color, &s.sheen_roughness, &s.transmission, &s.thickness};
        for (int k = 0; k < SLOT_COUNT; k++) {
            const pt_texture_sample& a = *slots[k];
            RTex& t = r.tex[k];
            if (a.descriptor == -1) {                     // unbound: a 1x1 white texel keeps the batched fetch branch-free
                t.texels = ctx->d_white; t.width = 1; t.height = 1; t.flags = RT_POINT;
                t.m00 = 0; t.m01 = 0; t.ox = 0; t.m10 = 0; t.m11 = 0; t.oy = 0;
                continue;
            }
            const TextureRec& tx = ctx->textures[a.descriptor];
            const SamplerRec& sm = ctx->samplers[a.sampler];
            float sn = 0.0f, cs = 1.0f;                   // sin(0) = 0, cos(0) = 1 exactly
            if (a.rotation != 0.0f) { sn = sinf(a.rotation); cs = cosf(a.rotation); }
            t.texels = tx.texels; t.width = tx.width; t.height = tx.height;
            t.flags = (tx.srgb ? RT_SRGB : 0u) | ((uint32_t)sm.address_u << 1) | ((uint32_t)sm.address_v << 3) |
                      (sm.mag_filter == PT_FILTER_POINT ? RT_POINT : 0u) | ((a.tex_coord & 1) ? RT_TEXCOORD1 : 0u);
            t.m00 = cs * a.scale[0]; t.m01 = sn * a.scale[1]; t.ox = a.offset[0];
            t.m10 = -sn * a.scale[0]; t.m11 = cs * a.scale[1]; t.oy = a.offset[1];
            r.bound_mask |= 1u << k;
        }
    }
    HIPOK(upload_table(ctx->d_rmats, ctx->rmats_cap, rm, ctx->stream));
    ctx->n_materials = count;
    return PT_OK;
}

int pt_scene_set_lights(pt_ctx* ctx, const pt_light* l, int count) {
    if (!ctx || (count > 0 && !l) || count < 0) return PT_ERR_INVALID_ARGUMENT;
    std::vector<pt_light> h(l, l + count);
    HIPOK(upload_table(ctx->d_lights, ctx->lights_cap, h, ctx->stream));
    ctx->n_lights = count;
    return PT_OK;
}

int pt_scene_set_instances(pt_ctx* ctx, const pt_instance_desc* in, int count) {
    if (!ctx || (count > 0 && !in) || count < 0) return PT_ERR_INVALID_ARGUMENT;
    if (count > PT_MAX_TLAS_INSTANCES) return ctx->fail(PT_ERR_CAPACITY, "more than PT_MAX_TLAS_INSTANCES instances");   // RayTracingAccelerationStructure.cpp:294-297
    std::vector<InstanceRec> recs;
    uint64_t tris = 0;
    auto check = [&](int h, int fmt_a, int fmt_b, size_t count_needed) -> bool {
        if (h == -1) return true;
        if (h < 0 || h >= (int)ctx->buffers.size()) return false;
        const BufferRec& b = ctx->buffers[h];
        if ((int)b.format != fmt_a && (int)b.format != fmt_b) return false;
        return b.bytes >= count_needed * format_stride(b.format);
    };
    for (int i = 0; i < count; i++) {
        const pt_instance_desc& d = in[i];
        const pt_mesh_instance& g = d.gpu;
        if (g.position_descriptor == -1 || !check(g.position_descriptor, PT_FORMAT_R32G32B32_FLOAT, PT_FORMAT_R32G32B32_FLOAT, d.num_of_vertices))
            return ctx->fail(PT_ERR_BAD_HANDLE, "instance position stream");
        if (!check(g.index_descriptor, PT_FORMAT_R16_UINT, PT_FORMAT_R32_UINT, d.num_of_indices)) return ctx->fail(PT_ERR_BAD_HANDLE, "instance index stream");
        if (g.index_descriptor == -1 && d.num_of_indices > d.num_of_vertices) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "non-indexed instance: num_of_indices > num_of_vertices");
        if (!check(g.tangent_space_descriptor, PT_FORMAT_R10G10B10A2_UNORM, PT_FORMAT_R10G10B10A2_UNORM, d.num_of_vertices)) return ctx->fail(PT_ERR_BAD_HANDLE, "instance tangent-space stream");
        if (!check(g.texcoord_descriptors[0], PT_FORMAT_R32G32_FLOAT, PT_FORMAT_R32G32_FLOAT, d.num_of_vertices) ||
            !check(g.texcoord_descriptors[1], PT_FORMAT_R32G32_FLOAT, PT_FORMAT_R32G32_FLOAT, d.num_of_vertices))
            return ctx->fail(PT_ERR_BAD_HANDLE, "instance texcoord stream");
        if (!check(g.color_descriptor, PT_FORMAT_R16G16B16A16_UNORM, PT_FORMAT_R16G16B16A16_UNORM, d.num_of_vertices)) return ctx->fail(PT_ERR_BAD_HANDLE, "instance colour stream");
        if (g.material_id < 0 || g.material_id >= ctx->n_materials) return ctx->fail(PT_ERR_BAD_HANDLE, "instance material_id out of range (set materials first)");
        InstanceRec r;
        memset(&r, 0, sizeof(r));
        r.gpu = g;
        const float* M = g.transform;
        double det = (double)M[0] * ((double)M[5] * M[10] - (double)M[9] * M[6]) - (double)M[4] * ((double)M[1] * M[10] - (double)M[9] * M[2]) +
                     (double)M[8] * ((double)M[1] * M[6] - (double)M[5] * M[2]);
        r.mask_flags = (d.instance_mask & 0xffu) | ((d.instance_flags & PT_INSTANCE_FLAG_TRIANGLE_CULL_DISABLE) ? TF_CULL_DISABLE : 0u) |
                       ((d.instance_flags & PT_INSTANCE_FLAG_FORCE_NON_OPAQUE) ? TF_FORCE_NON_OPAQUE : 0u) | (det < 0 ? TF_MIRRORED : 0u);
        auto ptr_of = [&](int h) -> const void* { return h == -1 ? nullptr : ctx->buffers[h].ptr; };
        r.p_index = ptr_of(g.index_descriptor);
        r.index_is16 = g.index_descriptor != -1 && ctx->buffers[g.index_descriptor].format == PT_FORMAT_R16_UINT;
        r.p_position = (const float*)ptr_of(g.position_descriptor);
        r.p_tangent_space = (const uint32_t*)ptr_of(g.tangent_space_descriptor);
        r.p_texcoord[0] = (const float2*)ptr_of(g.texcoord_descriptors[0]);
        r.p_texcoord[1] = (const float2*)ptr_of(g.texcoord_descriptors[1]);
        r.p_color = (const uint2*)ptr_of(g.color_descriptor);
        r.tri_offset = (uint32_t)tris;
        r.tri_count = d.num_of_indices / 3;
        tris += r.tri_count;
        if (tris > 0x0fffffffull) return ctx->fail(PT_ERR_CAPACITY, "too many triangles (leaf references hold 28 bits of triangle index)");
        recs.push_back(r);
    }
    ctx->instances.swap(recs);
    ctx->n_tris = (uint32_t)tris;
    ctx->accel_dirty = true;
    return PT_OK;
}

int pt_env_create(pt_ctx* ctx, const float* rgb, int width, int height, int* env_out) {
    if (!ctx || !rgb || !env_out || width <= 0 || height <= 0) return PT_ERR_INVALID_ARGUMENT;
    float* d = nullptr;
    size_t bytes = (size_t)width * height * 12;
    HIPOK(hipMalloc((void**)&d, bytes));
    hipError_t e = hipMemcpy(d, rgb, bytes, hipMemcpyHostToDevice);
    EnvDevice* env = new EnvDevice();
    if (!e) e = env_build(*env, d, width, height, ctx->stream);
    if (!e) e = hipStreamSynchronize(ctx->stream);
    hipFree(d);
    if (e) { env_free(*env); delete env; return ctx->fail(PT_ERR_DEVICE, std::string("pt_env_create: ") + hipGetErrorString(e)); }
    ctx->envs.push_back(env);
    *env_out = (int)ctx->envs.size() - 1;
    return PT_OK;
}

int pt_env_read(pt_ctx* ctx, int env, int* cube_size_out, uint16_t* cube, float* pyramid) {
    if (!ctx) return PT_ERR_INVALID_ARGUMENT;
    if (env < 0 || env >= (int)ctx->envs.size()) return ctx->fail(PT_ERR_BAD_HANDLE, "pt_env_read");
    HIPOK(hipStreamSynchronize(ctx->stream));
    const EnvDevice& ed = *ctx->envs[env];
    int n = ed.mip_n[0];
    size_t pyr = 0;
    for (int r = ed.imp_res; r >= 1; r >>= 1) pyr += (size_t)r * r;
    const uint16_t* dc = ed.cube; const float* dp = ed.importance;
    if (cube_size_out) *cube_size_out = n;
    if (cube) HIPOK(hipMemcpy(cube, dc, (size_t)6 * n * n * 4 * 2, hipMemcpyDeviceToHost));
    if (pyramid) HIPOK(hipMemcpy(pyramid, dp, pyr * 4, hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_build_accel(pt_ctx* ctx) {
    if (!ctx) return PT_ERR_INVALID_ARGUMENT;
    return Pathtracer::BuildAccel(ctx);
}

int pt_skin_run(pt_ctx* ctx, const pt_skin_params* params, const pt_bone* bones, int bone_count) {
    if (!ctx || !params) return PT_ERR_INVALID_ARGUMENT;
    return GpuSkin::Run(ctx, params, bones, bone_count);
}

int pt_trace(pt_ctx* ctx, const pt_settings* settings, const pt_execute_params* params) {
    if (!ctx || !settings || !params) return PT_ERR_INVALID_ARGUMENT;
    if (params->width == 0 || params->height == 0) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "zero resolution");
    return Pathtracer::PathtraceScene(ctx, settings, params);
}

int pt_set_bounce_limit(pt_ctx* ctx, int limit) {
    if (!ctx || limit < 0) return PT_ERR_INVALID_ARGUMENT;
    ctx->bounce_limit = limit;
    return PT_OK;
}

int pt_set_samples_per_trace(pt_ctx* ctx, int samples) {
    if (!ctx || samples < 1 || samples > PT_MAX_SAMPLES_PER_TRACE) return PT_ERR_INVALID_ARGUMENT;
    ctx->samples_per_trace = samples;
    return PT_OK;
}

int pt_set_null_shadow_culling(pt_ctx* ctx, int enable) {
    if (!ctx) return PT_ERR_INVALID_ARGUMENT;
    ctx->cull_null_shadow = enable != 0;
    return PT_OK;
}

int pt_set_kernel_mode(pt_ctx* ctx, int mode, int stage_blocks) {
    if (!ctx || (mode != PT_MODE_WAVEFRONT && mode != PT_MODE_MEGAKERNEL)) return PT_ERR_INVALID_ARGUMENT;
    ctx->kernel_mode = mode;
    if (stage_blocks > 0) ctx->stage_blocks = stage_blocks;
    return PT_OK;
}

int pt_enable_counters(pt_ctx* ctx, int enable) {
    if (!ctx) return PT_ERR_INVALID_ARGUMENT;
    ctx->counters_enabled = enable != 0;
    return PT_OK;
}

int pt_reset_stats(pt_ctx* ctx) {
    if (!ctx) return PT_ERR_INVALID_ARGUMENT;
    HIPOK(hipMemsetAsync(ctx->d_counters, 0, sizeof(Counters), ctx->stream));
    return PT_OK;
}

int pt_get_stats(pt_ctx* ctx, pt_stats* out) {
    if (!ctx || !out) return PT_ERR_INVALID_ARGUMENT;
    HIPOK(hipStreamSynchronize(ctx->stream));
    Counters c;
    HIPOK(hipMemcpy(&c, ctx->d_counters, sizeof(c), hipMemcpyDeviceToHost));
    memset(out, 0, sizeof(*out));
    out->rays_primary = c.rays_primary; out->rays_bounce = c.rays_bounce; out->rays_shadow = c.rays_shadow;
    out->rays = c.rays_primary + c.rays_bounce + c.rays_shadow;
    out->nodes_visited = c.nodes; out->tris_tested = c.tris; out->closest_hits = c.hits; out->texture_taps = c.taps;
    if (ctx->have_trace) hipEventElapsedTime(&out->trace_ms, ctx->ev_trace[0], ctx->ev_trace[1]);
    if (ctx->have_accel) hipEventElapsedTime(&out->accel_ms, ctx->ev_accel[0], ctx->ev_accel[1]);
    if (ctx->have_skin) hipEventElapsedTime(&out->skin_ms, ctx->ev_skin[0], ctx->ev_skin[1]);
    out->accumulated_frames = ctx->accumulated_frames;
    if (ctx->wide_nodes == kWideNodesOnDevice)      // small scene: the build did not wait for the count (pt_host.h)
        HIPOK(hipMemcpy(&ctx->wide_nodes, ctx->scratch.collapse_counters, 4, hipMemcpyDeviceToHost));
    out->bvh_nodes = ctx->wide_nodes;
    out->bvh_triangles = ctx->n_tris;
    if (c.stack_overflow) return ctx->fail(PT_ERR_CAPACITY, "traversal stack overflow: " + std::to_string(c.stack_overflow) + " pushes dropped");
    return PT_OK;
}

int pt_readback(pt_ctx* ctx, const void* device_rgba32f, uint32_t width, uint32_t height, float* host) {
    if (!ctx || !device_rgba32f || !host) return PT_ERR_INVALID_ARGUMENT;
    HIPOK(hipStreamSynchronize(ctx->stream));
    HIPOK(hipMemcpy(host, device_rgba32f, (size_t)width * height * 16, hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_tonemap(pt_ctx* ctx, const pt_tonemap_config* cfg, const void* device_rgba32f, uint32_t width, uint32_t height, float* host_rgb, uint8_t* host_rgba8) {
    if (!ctx || !cfg || !device_rgba32f || width == 0 || height == 0) return PT_ERR_INVALID_ARGUMENT;
    size_t n = (size_t)width * height;
    float* d_rgb = nullptr; uint32_t* d_q = nullptr;
    if (host_rgb) HIPOK(hipMalloc((void**)&d_rgb, n * 12));
    if (host_rgba8) HIPOK(hipMalloc((void**)&d_q, n * 4));
    launch_tonemap((const float4*)device_rgba32f, width, height, *cfg, d_rgb, d_q, ctx->stream);
    hipError_t e = hipGetLastError();
    if (!e) e = hipStreamSynchronize(ctx->stream);
    if (!e && host_rgb) e = hipMemcpy(host_rgb, d_rgb, n * 12, hipMemcpyDeviceToHost);
    if (!e && host_rgba8) e = hipMemcpy(host_rgba8, d_q, n * 4, hipMemcpyDeviceToHost);
    hipFree(d_rgb); hipFree(d_q);
    if (e) return ctx->fail(PT_ERR_DEVICE, std::string("pt_tonemap: ") + hipGetErrorString(e));
    return PT_OK;
}

}  // extern "C"
