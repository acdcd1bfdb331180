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
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "pt_types.h"

#include "pt_host.h"

using namespace pt;

// Pinned upload slots for the per-frame tables (materials, lights, instances, bones, vertex updates).  The reference hands those
// over in a transient upload heap that stays valid for the frame (Source/Renderer.cpp:490,496); here the caller's memory may be
// reused as soon as the call returns, so the bytes are copied into a pinned slot and go to the device asynchronously.  The host
// only ever waits when it comes round to a slot whose copy is still in flight -- not once per call.
struct StagingRing {
    static constexpr int kSlots = 8;
    void* host[kSlots] = {};
    size_t cap[kSlots] = {};
    hipEvent_t done[kSlots] = {};
    bool pending[kSlots] = {};
    int next = 0;
};

enum AccelState { ACCEL_CLEAN = 0, ACCEL_REFIT = 1, ACCEL_REBUILD = 2 };


struct pt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string error;
    StagingRing staging;

    // ---- ResourceTable ("descriptor heap")
    std::vector<BufferRec> buffers;
    std::vector<TextureRec> textures;
    std::vector<SamplerRec> samplers;
    BufferRec* d_buffers = nullptr; size_t d_buffers_cap = 0; bool buffers_dirty = true;   // device copy: BVH build only
    uint32_t* d_white = nullptr;                                   // 1x1 white texel behind every unbound material slot
    // interleaved albedo / normal / metal-rough texels of the materials whose three textures share one footprint (pt_types.h RM_TRIO),
    // keyed by the three texel pointers (nullptr = slot unbound); owned here, rebuilt / released by pt_scene_set_materials
    struct TrioRec { const uint32_t *a, *n, *m, *e; uint4* ptr; };
    std::vector<TrioRec> trios;

    // ---- per-frame arrays (Renderer::GatherMaterials / GatherLights)
    RMat* d_rmats = nullptr; int n_materials = 0; size_t rmats_cap = 0;    // resolved on the host in pt_scene_set_materials
    std::vector<RMat> rmats_host;                                          // what was uploaded (pt_texture_destroy checks it)
    pt_light* d_lights = nullptr; int n_lights = 0; size_t lights_cap = 0;

    // ---- instance table + acceleration structure
    std::vector<InstanceRec> instances;
    InstanceRec* d_instances = nullptr; size_t instances_cap = 0;
    uint32_t n_tris = 0;
    Bvh4Node* d_nodes = nullptr; TriPacket* d_tris = nullptr; ShadePacket* d_shade = nullptr; size_t accel_cap = 0;
    uint32_t wide_nodes = 0, stack_need = 0;
    int32_t root = 0;
    AccelScratch scratch;
    // What the next pt_build_accel / pt_trace has to do: nothing, a refit of the instances marked in `touched` (vertices or
    // transform changed: UpdateDynamicBlas + the per-frame TLAS rebuild upstream), or a full build (topology changed).
    int accel_state = ACCEL_REBUILD;
    bool accel_built = false;                 // a full build of the current instance table exists (a refit needs one)
    bool instances_dirty = true;              // the device copy of the instance table is stale
    std::vector<uint8_t> touched;             // per instance
    uint8_t* d_touched = nullptr; size_t touched_cap = 0;
    uint32_t accel_refits = 0, accel_builds = 0;
    std::vector<int> free_buffers, free_textures, free_envs;      // destroyed handles, reused by the next create

    std::vector<EnvDevice*> envs;
    float* d_sheen = nullptr;
    float* d_srgb = nullptr;
    float2* d_tangent_lut = nullptr;
    Counters* d_counters = nullptr;
    void* d_bones = nullptr; size_t bones_cap = 0, bones_used = 0;   // bone arena: one slice per pt_skin_run, wraps behind a fence
    hipEvent_t bones_fence = nullptr; bool bones_fence_pending = false;
    void* d_workspace = nullptr; size_t workspace_cap = 0;     // wavefront ray / hit / path-state arrays
    void* d_tonemap = nullptr; size_t tonemap_cap = 0;         // pt_tonemap's device scratch (float RGB + RGBA8), reused
    pt::ExchangeState* exchange = nullptr;                         // pt_exchange_* (exchange.hip)
    int32_t* d_deep = nullptr; size_t deep_cap = 0;                // deep traversal stack (SceneRec::deep_stack), only for trees that need > 64 entries
    uint32_t* d_occ = nullptr; size_t occ_pixels = 0; uint32_t occ_w = 0, occ_h = 0; bool occ_stale = true;   // occluder cache (pt_wavefront.hip), per output resolution
    bool occ_enabled = false;                                       // only with a -DPT_OCC_CACHE=1 build and MIPT_OCC_CACHE=1: measured, no gain (pt_wavefront.hip)
    bool stage_timing = false;                                 // pt_enable_stage_timing
    StageTimers timers;
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
// Every entry point that touches the device makes the context's device current first: two contexts on two GPUs in one process
// (one per rank thread, or a host that drives several GPUs itself) must not launch or allocate on each other's device.
#define ENTER(ctx)                                                                                           \
    do {                                                                                                     \
        hipError_t _e = hipSetDevice((ctx)->device);                                                         \
        if (_e != hipSuccess) return (ctx)->fail(PT_ERR_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(_e)); \
    } while (0)

namespace {

// Host -> device through a pinned slot of the ring, asynchronous on the context's stream.
hipError_t staged_upload(pt_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (bytes == 0) return hipSuccess;
    StagingRing& r = ctx->staging;
    const int k = r.next;
    r.next = (r.next + 1) % StagingRing::kSlots;
    hipError_t e;
    if (r.pending[k]) { if ((e = hipEventSynchronize(r.done[k]))) return e; r.pending[k] = false; }
    if (bytes > r.cap[k]) {
        if (r.host[k]) hipHostFree(r.host[k]);
        r.host[k] = nullptr; r.cap[k] = 0;
        const size_t nc = bytes + bytes / 2 + 4096;
        if ((e = hipHostMalloc(&r.host[k], nc, hipHostMallocDefault))) return e;
        r.cap[k] = nc;
    }
    if (!r.done[k] && (e = hipEventCreateWithFlags(&r.done[k], hipEventDisableTiming))) return e;
    memcpy(r.host[k], src, bytes);
    if ((e = hipMemcpyAsync(dst, r.host[k], bytes, hipMemcpyHostToDevice, ctx->stream))) return e;
    if ((e = hipEventRecord(r.done[k], ctx->stream))) return e;
    r.pending[k] = true;
    return hipSuccess;
}

template <typename T>
hipError_t upload_table(pt_ctx* ctx, T*& d, size_t& cap, const std::vector<T>& h) {
    size_t n = h.size() ? h.size() : 1;
    if (n > cap) {
        // kernels already enqueued may still read the old table
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e) return e;
        hipFree(d);
        d = nullptr; cap = 0;
        size_t nc = n + n / 2 + 8;
        e = hipMalloc((void**)&d, nc * sizeof(T));
        if (e) return e;
        cap = nc;
    }
    if (h.empty()) return hipSuccess;
    return staged_upload(ctx, d, h.data(), h.size() * sizeof(T));
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

// A vertex stream was rewritten (pt_skin_run, pt_buffer_update): the instances that read it need their packets rebuilt (refit);
// a rewritten index stream changes which vertices form a triangle, which the refit also handles (it re-reads the indices), but
// the Morton order was made for the old triangles -- still correct, only slower -- so that stays a refit too.
void mark_buffer_users(pt_ctx* ctx, int handle) {
    if (handle < 0) return;
    if (ctx->touched.size() != ctx->instances.size()) ctx->touched.assign(ctx->instances.size(), 0);
    const void* ptr = ctx->buffers[handle].ptr;
    bool any = false;
    for (size_t i = 0; i < ctx->instances.size(); i++) {
        const InstanceRec& r = ctx->instances[i];
        if (r.p_index == ptr || r.p_position == ptr || r.p_tangent_space == ptr || r.p_texcoord[0] == ptr || r.p_texcoord[1] == ptr || r.p_color == ptr) {
            ctx->touched[i] = 1;
            any = true;
        }
    }
    if (any && ctx->accel_state == ACCEL_CLEAN) ctx->accel_state = ACCEL_REFIT;
}

}  // namespace

// =================================================================================================
// class Pathtracer (Source/Pathtracer.h:16-157)
namespace pt {
class Pathtracer {
public:
    // Pathtracer::BuildAllBlas + UpdateAllBlas + BuildTlas (Source/Pathtracer.cpp:138-257).  Upstream builds a BLAS once per
    // primitive, refits the dynamic ones every frame (UpdateDynamicBlas, RayTracingAccelerationStructure.cpp:110-158) and rebuilds a
    // small TLAS every frame.  Here: ONE full build of the flattened soup when the set of triangles changes, and a refit -- packets
    // of the touched instances rewritten in place, every box re-derived -- when only vertices or transforms moved.
    static int BuildAccel(pt_ctx* ctx) {
        if (ctx->buffers_dirty) { HIPOK(upload_table(ctx, ctx->d_buffers, ctx->d_buffers_cap, ctx->buffers)); ctx->buffers_dirty = false; }
        if (ctx->instances_dirty) { HIPOK(upload_table(ctx, ctx->d_instances, ctx->instances_cap, ctx->instances)); ctx->instances_dirty = false; }
        if (ctx->accel_state == ACCEL_CLEAN && ctx->accel_built) return PT_OK;
        const bool refit = ctx->accel_built && ctx->accel_state == ACCEL_REFIT;
        if (!refit) {
            size_t need = ctx->n_tris ? ctx->n_tris : 1;
            if (need > ctx->accel_cap) {
                HIPOK(hipStreamSynchronize(ctx->stream));
                hipFree(ctx->d_nodes); hipFree(ctx->d_tris); hipFree(ctx->d_shade);
                ctx->d_nodes = nullptr; ctx->d_tris = nullptr; ctx->d_shade = nullptr; ctx->accel_cap = 0;
                size_t cap = need + need / 8 + 64;
                HIPOK(hipMalloc((void**)&ctx->d_nodes, cap * sizeof(Bvh4Node)));
                HIPOK(hipMalloc((void**)&ctx->d_tris, cap * sizeof(TriPacket)));
                HIPOK(hipMalloc((void**)&ctx->d_shade, cap * sizeof(ShadePacket)));
                ctx->accel_cap = cap;
            }
        }
        HIPOK(hipEventRecord(ctx->ev_accel[0], ctx->stream));
        if (refit) {
            const size_t n = ctx->instances.size();
            if (n > ctx->touched_cap) {
                HIPOK(hipStreamSynchronize(ctx->stream));
                hipFree(ctx->d_touched); ctx->d_touched = nullptr; ctx->touched_cap = 0;
                HIPOK(hipMalloc((void**)&ctx->d_touched, n + 64));
                ctx->touched_cap = n + 64;
            }
            HIPOK(staged_upload(ctx, ctx->d_touched, ctx->touched.data(), n));
            HIPOK(accel_refit(ctx->scratch, ctx->d_instances, ctx->d_touched, ctx->n_tris, ctx->wide_nodes, ctx->d_nodes, ctx->d_tris, ctx->d_shade, ctx->stream));
            ctx->accel_refits++;
        } else {
            ctx->scratch.why.clear();
            const hipError_t be = accel_build(ctx->scratch, ctx->d_buffers, ctx->d_instances, (int)ctx->instances.size(), ctx->n_tris, ctx->d_nodes, ctx->d_tris,
                                              ctx->d_shade, &ctx->root, &ctx->wide_nodes, &ctx->stack_need, ctx->stream);
            if (be != hipSuccess) {
                ctx->accel_built = false; ctx->accel_state = ACCEL_REBUILD;
                return ctx->fail(PT_ERR_DEVICE, std::string("acceleration-structure build: ") + (ctx->scratch.why.empty() ? hipGetErrorString(be) : ctx->scratch.why.c_str()));
            }
            ctx->accel_builds++;
            ctx->occ_stale = true;                 // triangle indices changed: the occluder cache's hints point at other triangles now
        }
        HIPOK(hipEventRecord(ctx->ev_accel[1], ctx->stream));
        ctx->have_accel = true;
        ctx->accel_state = ACCEL_CLEAN;
        ctx->accel_built = true;
        std::fill(ctx->touched.begin(), ctx->touched.end(), (uint8_t)0);
        // The build reports the most stack entries any ray can hold in this tree (a node pushes its other children).  A lane holds 64
        // on chip (LDS + scratch).  A tree that needs more -- long chains of coincident centroids; the reference's driver BVH logs and
        // skips only a primitive it cannot build, RayTracingAccelerationStructure.cpp:137-140 -- is never refused and never drops pushes:
        // its rays get a deep stack in memory for the entries beyond 64 (SceneRec::deep_stack, sized at the next pt_trace).  Only a
        // clustered tree deeper than kDeepStackMax entries is rebuilt as a radix tree, whose depth is bounded by the key (64 Morton
        // bits + 32 index bits: at most 3 x 96 entries).  A refit keeps the topology, so the figure of the build stands.
        if (!refit && ctx->stack_need > kDeepStackMax && ctx->scratch.builder != 0) {
            const int chosen = ctx->scratch.builder;
            ctx->scratch.builder = 0;
            ctx->scratch.fallbacks++;
            ctx->scratch.fallback_why = "clustered tree needs a traversal stack of " + std::to_string(ctx->stack_need) + " entries";
            const hipError_t be = accel_build(ctx->scratch, ctx->d_buffers, ctx->d_instances, (int)ctx->instances.size(), ctx->n_tris, ctx->d_nodes, ctx->d_tris,
                                              ctx->d_shade, &ctx->root, &ctx->wide_nodes, &ctx->stack_need, ctx->stream);
            ctx->scratch.builder = chosen;
            if (be != hipSuccess) {
                ctx->accel_built = false; ctx->accel_state = ACCEL_REBUILD;
                return ctx->fail(PT_ERR_DEVICE, std::string("acceleration-structure build (radix fallback): ") + (ctx->scratch.why.empty() ? hipGetErrorString(be) : ctx->scratch.why.c_str()));
            }
            HIPOK(hipEventRecord(ctx->ev_accel[1], ctx->stream));
        }
        if (!refit && ctx->stack_need > kDeepStackMax) {
            ctx->accel_built = false; ctx->accel_state = ACCEL_REBUILD;
            return ctx->fail(PT_ERR_CAPACITY, "acceleration structure needs a traversal stack of " + std::to_string(ctx->stack_need) + " entries (limit " +
                                              std::to_string(kDeepStackMax) + ")");
        }
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
            if (ep->environment_map >= 0 && (ep->environment_map >= (int)ctx->envs.size() || !ctx->envs[ep->environment_map]))
                return ctx->fail(PT_ERR_BAD_HANDLE, "bad environment map handle");
            if (!ep->output) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "output is null");
            if (ctx->accel_state != ACCEL_CLEAN || !ctx->accel_built || ctx->instances_dirty) { int r = BuildAccel(ctx); if (r) return r; }

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
            {   // a sheen material among at most a tenth of the materials: its hits are shaded together (the default; MIPT_DEFER_RARE=0 / 1 forces it)
                size_t rare = 0;
                for (const RMat& m : ctx->rmats_host) rare += (m.sheen_color_factor[0] != 0.0f || m.sheen_color_factor[1] != 0.0f || m.sheen_color_factor[2] != 0.0f) ? 1 : 0;
                fc.defer_rare = (rare != 0 && rare * 10 <= ctx->rmats_host.size()) ? 1u : 0u;
                static const int forced = [] { const char* e = getenv("MIPT_DEFER_RARE"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }();
                if (forced >= 0) fc.defer_rare = (uint32_t)forced;
            }
            fc.spp = 1; fc.pixel_slots = fc.my_tiles * 256u;
            fc.div_pixel_slots = FastDiv::make(fc.pixel_slots); fc.div_tiles_x = FastDiv::make(fc.tiles_x);
            fc.seed_step = settings->use_frame_as_seed ? 1u : 0u;

            // deep traversal stack (trees that need more than the 64 on-chip entries): (need - 64) entries for every lane of the widest
            // traversal launch of this call
            if ((int)ctx->stack_need > traversal_stack_capacity()) {
                const uint32_t entries = (ctx->stack_need - (uint32_t)traversal_stack_capacity() + 7u) & ~7u;
                size_t lanes;
                if (ctx->kernel_mode == PT_MODE_MEGAKERNEL) lanes = (size_t)fc.my_tiles * 256;
                else lanes = (size_t)traversal_grid_lanes(ctx->stage_blocks > 0 ? ctx->stage_blocks : stage_blocks_for((size_t)fc.pixel_slots * (size_t)batch));
                const size_t need = (size_t)entries * lanes * 4;
                if (need > ctx->deep_cap) {
                    HIPOK(hipStreamSynchronize(ctx->stream));
                    hipFree(ctx->d_deep); ctx->d_deep = nullptr; ctx->deep_cap = 0;
                    if (hipMalloc((void**)&ctx->d_deep, need) != hipSuccess) { (void)hipGetLastError(); return ctx->fail(PT_ERR_OUT_OF_MEMORY, "deep traversal stack: " + std::to_string(need) + " bytes"); }
                    ctx->deep_cap = need;
                }
                sc.deep_stack = ctx->d_deep; sc.deep_entries = entries; sc.deep_lanes = (uint32_t)lanes;
            }
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
                size_t need = wavefront_workspace_bytes(fc, stage_blocks);
                if (need > ctx->workspace_cap) {
                    HIPOK(hipStreamSynchronize(ctx->stream));
                    hipFree(ctx->d_workspace); ctx->d_workspace = nullptr; ctx->workspace_cap = 0;
                    HIPOK(hipMalloc(&ctx->d_workspace, need));
                    ctx->workspace_cap = need;
                }
                // the occluder cache follows the output resolution; triangle indices change with every full build (not with a refit), which
                // makes its content stale: hints are re-validated by an exact test anyway, but a clean table costs one memset
                uint32_t* occ = nullptr;
                if (ctx->occ_enabled && settings->debug_output == PT_DEBUG_OUTPUT_NONE) {
                    const size_t px = (size_t)ep->width * ep->height;
                    if (px != ctx->occ_pixels || ep->width != ctx->occ_w) {
                        HIPOK(hipStreamSynchronize(ctx->stream));
                        hipFree(ctx->d_occ); ctx->d_occ = nullptr; ctx->occ_pixels = 0;
                        if (hipMalloc((void**)&ctx->d_occ, px * 8 * 4) == hipSuccess) { ctx->occ_pixels = px; ctx->occ_w = ep->width; ctx->occ_h = ep->height; ctx->occ_stale = true; }
                        else (void)hipGetLastError();
                    }
                    if (ctx->d_occ && ctx->occ_stale) { HIPOK(hipMemsetAsync(ctx->d_occ, 0xff, ctx->occ_pixels * 8 * 4, ctx->stream)); ctx->occ_stale = false; }
                    occ = ctx->d_occ;
                }
                HIPOK(launch_wavefront(sc, fc, (float4*)ep->output, ctx->d_counters, ctx->counters_enabled, ctx->d_workspace, stage_blocks,
                                       ctx->stage_timing ? &ctx->timers : nullptr, ctx->stream, occ));
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
            // Each call gets its own slice of the bone arena: a frame skins several primitives back to back on one stream, and the
            // kernel of the previous call may not have read its bones yet (they live in the caller's transient heap, Renderer.cpp:411).
            if (ctx->bones_used + bytes > ctx->bones_cap) {
                if (bytes > ctx->bones_cap) {
                    HIPOK(hipStreamSynchronize(ctx->stream));
                    hipFree(ctx->d_bones); ctx->d_bones = nullptr; ctx->bones_cap = 0;
                    const size_t nc = bytes * 8 + 4096;
                    HIPOK(hipMalloc(&ctx->d_bones, nc));
                    ctx->bones_cap = nc;
                } else if (ctx->bones_fence_pending) HIPOK(hipEventSynchronize(ctx->bones_fence));      // wrap: the arena's earlier readers must be done
                ctx->bones_used = 0;
            }
            void* dst = (char*)ctx->d_bones + ctx->bones_used;
            ctx->bones_used += (bytes + 255) & ~(size_t)255;
            HIPOK(staged_upload(ctx, dst, bones, bytes));
            a.bones = (const pt_bone*)dst;
            a.bone_count = bone_count;
        }
        HIPOK(hipEventRecord(ctx->ev_skin[0], ctx->stream));
        launch_skin(a, p->use_mfma != 0, ctx->stream);
        HIPOK(hipGetLastError());
        HIPOK(hipEventRecord(ctx->ev_skin[1], ctx->stream));
        if (a.bones) { HIPOK(hipEventRecord(ctx->bones_fence, ctx->stream)); ctx->bones_fence_pending = true; }
        ctx->have_skin = true;
        // UpdateAllBlas refits every dynamic primitive every frame (Pathtracer.cpp:168-183): the instances that read the streams
        // just written are refitted by the next pt_build_accel / pt_trace
        if (a.out_position) mark_buffer_users(ctx, p->output_position);
        if (a.out_tangent_space) mark_buffer_users(ctx, p->output_tangent_space);
        return PT_OK;
    }
};
}  // namespace pt

static int exchange_frame_checked(pt_ctx* ctx, const void* local, void* frame, uint32_t w, uint32_t h, int mode, int dst, std::string& err) {
    // (rank / world live in the exchange state; the root needs somewhere to put the frame)
    return exchange_frame(ctx->exchange, local, frame ? frame : const_cast<void*>(local), w, h, mode, dst, ctx->stream, err);
}

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
    ok = ok && hipEventCreateWithFlags(&ctx->bones_fence, hipEventDisableTiming) == hipSuccess;
    if (!ok) { pt_destroy(ctx); return PT_ERR_DEVICE; }
    ctx->samplers.push_back({PT_ADDRESS_WRAP, PT_ADDRESS_WRAP, PT_FILTER_LINEAR, PT_FILTER_LINEAR});   // sampler 0 (GpuResources.cpp:47-59)
    if (const char* b = getenv("MIPT_ACCEL_BUILDER")) {      // initial builder of new contexts (pt_set_accel_builder overrides): "lbvh" | "ploc" | "reinsert"
        if (!strcmp(b, "ploc")) ctx->scratch.builder = PT_BUILDER_PLOC;
        else if (!strcmp(b, "lbvh")) ctx->scratch.builder = PT_BUILDER_LBVH;
        else if (!strcmp(b, "reinsert")) ctx->scratch.builder = PT_BUILDER_PLOC_REINSERT;
    }
    if (const char* b = getenv("MIPT_OCC_CACHE")) ctx->occ_enabled = atoi(b) != 0;     // 0: no occluder cache (A/B runs; images are identical either way)
    if (const char* b = getenv("MIPT_REINSERT_PASSES")) {    // tuning aid (tools/builder_probe.py): passes of PT_BUILDER_PLOC_REINSERT
        const int v = atoi(b);
        if (v >= 0 && v <= 64) ctx->scratch.reinsert_passes = v;
    }
    if (const char* b = getenv("MIPT_REINSERT_MIN_GAIN")) {
        const float v = (float)atof(b);
        if (v >= 0.0f && v < 1.0f) ctx->scratch.reinsert_min_gain = v;
    }
    *out = ctx;
    return PT_OK;
}

void pt_destroy(pt_ctx* ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    for (auto& b : ctx->buffers) hipFree((void*)b.ptr);
    for (auto& t : ctx->textures) hipFree((void*)t.texels);
    for (auto& t : ctx->trios) hipFree((void*)t.ptr);
    for (auto* e : ctx->envs) if (e) { env_free(*e); delete e; }
    hipFree(ctx->d_buffers); hipFree(ctx->d_white); hipFree(ctx->d_rmats); hipFree(ctx->d_lights);
    hipFree(ctx->d_instances); hipFree(ctx->d_nodes); hipFree(ctx->d_tris); hipFree(ctx->d_shade); hipFree(ctx->d_sheen); hipFree(ctx->d_srgb); hipFree(ctx->d_tangent_lut); hipFree(ctx->d_counters);
    accel_scratch_free(ctx->scratch);
    hipFree(ctx->d_bones);
    hipFree(ctx->d_workspace);
    hipFree(ctx->d_tonemap);
    hipFree(ctx->d_touched);
    exchange_free(ctx->exchange);
    hipFree(ctx->d_deep);
    hipFree(ctx->d_occ);
    for (int k = 0; k < StagingRing::kSlots; k++) {
        if (ctx->staging.host[k]) hipHostFree(ctx->staging.host[k]);
        if (ctx->staging.done[k]) hipEventDestroy(ctx->staging.done[k]);
    }
    for (hipEvent_t e : ctx->timers.ev) hipEventDestroy(e);
    if (ctx->bones_fence) hipEventDestroy(ctx->bones_fence);
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
    ENTER(ctx);
    void* d = nullptr;
    HIPOK(hipMalloc(&d, bytes + 16));
    hipError_t e = host ? hipMemcpy(d, host, bytes, hipMemcpyHostToDevice) : hipMemset(d, 0, bytes);
    if (e) { hipFree(d); return ctx->fail(PT_ERR_DEVICE, std::string("pt_buffer_create: ") + hipGetErrorString(e)); }
    const BufferRec rec = {d, (uint32_t)format, (uint32_t)bytes};
    if (!ctx->free_buffers.empty()) { *handle_out = ctx->free_buffers.back(); ctx->free_buffers.pop_back(); ctx->buffers[*handle_out] = rec; }
    else { ctx->buffers.push_back(rec); *handle_out = (int)ctx->buffers.size() - 1; }
    ctx->buffers_dirty = true;
    return PT_OK;
}

static bool live_buffer(const pt_ctx* ctx, int h) { return h >= 0 && h < (int)ctx->buffers.size() && ctx->buffers[h].ptr != nullptr; }

int pt_buffer_update(pt_ctx* ctx, int handle, const void* host, size_t bytes) {
    if (!ctx || !host) return PT_ERR_INVALID_ARGUMENT;
    if (!live_buffer(ctx, handle) || bytes > ctx->buffers[handle].bytes) return ctx->fail(PT_ERR_BAD_HANDLE, "pt_buffer_update");
    ENTER(ctx);
    HIPOK(staged_upload(ctx, (void*)ctx->buffers[handle].ptr, host, bytes));
    mark_buffer_users(ctx, handle);
    return PT_OK;
}

int pt_buffer_read(pt_ctx* ctx, int handle, void* host, size_t bytes) {
    if (!ctx || !host) return PT_ERR_INVALID_ARGUMENT;
    if (!live_buffer(ctx, handle) || bytes > ctx->buffers[handle].bytes) return ctx->fail(PT_ERR_BAD_HANDLE, "pt_buffer_read");
    ENTER(ctx);
    HIPOK(hipStreamSynchronize(ctx->stream));
    HIPOK(hipMemcpy(host, ctx->buffers[handle].ptr, bytes, hipMemcpyDeviceToHost));
    return PT_OK;
}

// Gltf::Unload (Source/Gltf.cpp:123-157) destroys every mesh, texture and dynamic mesh of the old scene before the next is loaded
// (Source/Main.cpp:43-54).  A resource the current instance table / material table still points at cannot go: the host replaces
// those tables first (pt_scene_set_instances / pt_scene_set_materials with the new scene, or with count 0), as upstream's
// per-frame tables are rebuilt from the new scene.
int pt_buffer_destroy(pt_ctx* ctx, int handle) {
    if (!ctx) return PT_ERR_INVALID_ARGUMENT;
    if (!live_buffer(ctx, handle)) return ctx->fail(PT_ERR_BAD_HANDLE, "pt_buffer_destroy");
    const void* ptr = ctx->buffers[handle].ptr;
    for (const InstanceRec& r : ctx->instances)
        if (r.p_index == ptr || r.p_position == ptr || r.p_tangent_space == ptr || r.p_texcoord[0] == ptr || r.p_texcoord[1] == ptr || r.p_color == ptr)
            return ctx->fail(PT_ERR_NOT_READY, "pt_buffer_destroy: the buffer is used by the current instance table (replace it with pt_scene_set_instances first)");
    ENTER(ctx);
    HIPOK(hipStreamSynchronize(ctx->stream));                // enqueued kernels (skinning, an earlier trace) may still read or write it
    hipFree((void*)ptr);
    ctx->buffers[handle] = BufferRec{nullptr, 0u, 0u};
    ctx->free_buffers.push_back(handle);
    ctx->buffers_dirty = true;
    return PT_OK;
}

int pt_texture_create(pt_ctx* ctx, const uint8_t* rgba8, int width, int height, int srgb, int* handle_out) {
    if (!ctx || !rgba8 || !handle_out || width <= 0 || height <= 0) return PT_ERR_INVALID_ARGUMENT;
    ENTER(ctx);
    void* d = nullptr;
    size_t bytes = (size_t)width * height * 4;
    HIPOK(hipMalloc(&d, bytes + 4));                         // (+ 4: the sampler reads the two texels of a row as one 8-byte pair, pt_shading.h texture_taps)
    hipError_t e = hipMemcpy(d, rgba8, bytes, hipMemcpyHostToDevice);
    if (e) { hipFree(d); return ctx->fail(PT_ERR_DEVICE, std::string("pt_texture_create: ") + hipGetErrorString(e)); }
    const TextureRec rec = {(const uint32_t*)d, width, height, srgb ? 1u : 0u, 0u};
    if (!ctx->free_textures.empty()) { *handle_out = ctx->free_textures.back(); ctx->free_textures.pop_back(); ctx->textures[*handle_out] = rec; }
    else { ctx->textures.push_back(rec); *handle_out = (int)ctx->textures.size() - 1; }
    return PT_OK;
}

int pt_texture_destroy(pt_ctx* ctx, int handle) {
    if (!ctx) return PT_ERR_INVALID_ARGUMENT;
    if (handle < 0 || handle >= (int)ctx->textures.size() || !ctx->textures[handle].texels) return ctx->fail(PT_ERR_BAD_HANDLE, "pt_texture_destroy");
    for (const RMat& m : ctx->rmats_host)
        for (int k = 0; k < SLOT_COUNT; k++)
            if (m.tex[k].texels == ctx->textures[handle].texels)
                return ctx->fail(PT_ERR_NOT_READY, "pt_texture_destroy: the texture is used by the current material table (replace it with pt_scene_set_materials first)");
    ENTER(ctx);
    HIPOK(hipStreamSynchronize(ctx->stream));
    hipFree((void*)ctx->textures[handle].texels);
    ctx->textures[handle] = TextureRec{nullptr, 0, 0, 0u, 0u};
    ctx->free_textures.push_back(handle);
    return PT_OK;
}

int pt_sampler_create(pt_ctx* ctx, const pt_sampler_desc* d, int* handle_out) {
    if (!ctx || !d || !handle_out) return PT_ERR_INVALID_ARGUMENT;
    if (d->address_u < 0 || d->address_u > 2 || d->address_v < 0 || d->address_v > 2) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "pt_sampler_create: address mode");
    ctx->samplers.push_back({d->address_u, d->address_v, d->min_filter, d->mag_filter});
    *handle_out = (int)ctx->samplers.size() - 1;
    return PT_OK;
}

// {albedo, normal, metal-rough, -} per texel from the three RGBA8 images of one size (an unbound one reads as white, like d_white)
__global__ void k_trio_interleave(uint4* __restrict__ dst, const uint32_t* __restrict__ a, const uint32_t* __restrict__ n, const uint32_t* __restrict__ m,
                                  const uint32_t* __restrict__ e, size_t count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) dst[i] = make_uint4(a[i], n ? n[i] : 0xffffffffu, m ? m[i] : 0xffffffffu, e ? e[i] : 0xffffffffu);
}

int pt_scene_set_materials(pt_ctx* ctx, const pt_material* m, int count) {
    if (!ctx || (count > 0 && !m) || count < 0) return PT_ERR_INVALID_ARGUMENT;
    // the instance table's material ids index this table: a shorter table must not leave one dangling (device reads of rmats[] and
    // of the LDS material cache are not bounds-checked)
    for (const InstanceRec& r : ctx->instances)
        if (r.gpu.material_id >= count)
            return ctx->fail(PT_ERR_BAD_HANDLE, "pt_scene_set_materials: the current instance table uses material " + std::to_string(r.gpu.material_id) +
                                                ", the new table has " + std::to_string(count) + " (replace the instances first)");
    ENTER(ctx);
    for (int i = 0; i < count; i++) {
        const pt_texture_sample* slots[15] = {&m[i].normal, &m[i].albedo, &m[i].metallic_roughness, &m[i].occlusion, &m[i].emissive, &m[i].specular,
                                              &m[i].specular_color, &m[i].clearcoat, &m[i].clearcoat_roughness, &m[i].clearcoat_normal, &m[i].anisotropy,
                                              &m[i].sheen_color, &m[i].sheen_roughness, &m[i].transmission, &m[i].thickness};
        for (auto* s : slots) {
            if (s->descriptor < -1 || s->descriptor >= (int)ctx->textures.size() || (s->descriptor >= 0 && !ctx->textures[s->descriptor].texels))
                return ctx->fail(PT_ERR_BAD_HANDLE, "material texture descriptor out of range or destroyed");
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
        r.anisotropy_cos = (float)cos((double)s.anisotropy_rotation); r.anisotropy_sin = (float)sin((double)s.anisotropy_rotation);   // correctly rounded, like the kernels' (pt_math.h pt_sincos)
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
            if (a.rotation != 0.0f) { sn = (float)sin((double)a.rotation); cs = (float)cos((double)a.rotation); }
            t.texels = tx.texels; t.width = tx.width; t.height = tx.height;
            t.flags = (tx.srgb ? RT_SRGB : 0u) | ((uint32_t)sm.address_u << 1) | ((uint32_t)sm.address_v << 3) |
                      (sm.mag_filter == PT_FILTER_POINT ? RT_POINT : 0u) | ((a.tex_coord & 1) ? RT_TEXCOORD1 : 0u);
            t.m00 = cs * a.scale[0]; t.m01 = sn * a.scale[1]; t.ox = a.offset[0];
            t.m10 = -sn * a.scale[0]; t.m11 = cs * a.scale[1]; t.oy = a.offset[1];
            r.bound_mask |= 1u << k;
        }
    }
    // Interleaved footprint (RM_TRIO): the albedo texture plus whichever of the normal and metal-rough textures are bound, when they
    // have its size, sampler state, UV set and UV transform -- the three bilinear footprints are then the same four texels, and the
    // shade stage reads them from one 16-B-a-texel copy (two cache lines a hit instead of six).  Same texels, same weights: images are
    // bit-identical either way (MIPT_TEXTURE_INTERLEAVE=0 keeps every material on the general path; tested).
    const char* env_trio = getenv("MIPT_TEXTURE_INTERLEAVE");
    const bool use_trio = !(env_trio && env_trio[0] == '0');
    std::vector<char> trio_live(ctx->trios.size(), 0);
    for (int i = 0; i < count && use_trio; i++) {
        RMat& r = rm[i];
        const RTex& A = r.tex[SLOT_ALBEDO];
        const RTex& N = r.tex[SLOT_NORMAL];
        const RTex& M = r.tex[SLOT_METALLIC_ROUGHNESS];
        const RTex& E = r.tex[SLOT_EMISSIVE];
        const bool ba = (r.bound_mask >> SLOT_ALBEDO) & 1u, bn = (r.bound_mask >> SLOT_NORMAL) & 1u, bm = (r.bound_mask >> SLOT_METALLIC_ROUGHNESS) & 1u;
        auto same_footprint = [&](const RTex& t) {
            return t.width == A.width && t.height == A.height && ((t.flags ^ A.flags) & ~(uint32_t)RT_SRGB) == 0 && memcmp(&t.m00, &A.m00, 6 * sizeof(float)) == 0;
        };
        // the emissive texture joins when it has the footprint (it is fetched on its own otherwise); it does not make a copy worth building alone
        const bool be = ((r.bound_mask >> SLOT_EMISSIVE) & 1u) && same_footprint(E);
        if (!ba || !(bn || bm) || (bn && !same_footprint(N)) || (bm && !same_footprint(M))) continue;
        const uint32_t* kn = bn ? N.texels : nullptr;
        const uint32_t* km = bm ? M.texels : nullptr;
        const uint32_t* ke = be ? E.texels : nullptr;
        size_t at = ctx->trios.size();
        for (size_t k = 0; k < ctx->trios.size(); k++)
            if (ctx->trios[k].a == A.texels && ctx->trios[k].n == kn && ctx->trios[k].m == km && ctx->trios[k].e == ke) { at = k; break; }
        if (at == ctx->trios.size()) {
            const size_t texels = (size_t)A.width * A.height;
            uint4* d = nullptr;
            if (hipMalloc((void**)&d, texels * 16 + 32) != hipSuccess) { (void)hipGetLastError(); continue; }     // no room: this material stays on the general path
            hipLaunchKernelGGL(k_trio_interleave, dim3((unsigned)((texels + 255) / 256)), dim3(256), 0, ctx->stream, d, A.texels, kn, km, ke, texels);
            HIPOK(hipMemsetAsync((char*)d + texels * 16, 0xff, 32, ctx->stream));
            ctx->trios.push_back({A.texels, kn, km, ke, d});
            trio_live.push_back(0);
        }
        trio_live[at] = 1;
        r.trio = ctx->trios[at].ptr;
        r.bound_mask |= RM_TRIO | ((bn && (N.flags & RT_SRGB)) ? RM_TRIO_SRGB_N : 0u) | ((bm && (M.flags & RT_SRGB)) ? RM_TRIO_SRGB_M : 0u) |
                        (be ? (RM_TRIO_EMISSIVE | ((E.flags & RT_SRGB) ? RM_TRIO_SRGB_E : 0u)) : 0u);
    }
    HIPOK(upload_table(ctx, ctx->d_rmats, ctx->rmats_cap, rm));
    ctx->rmats_host.swap(rm);
    ctx->n_materials = count;
    // copies the new table no longer names: enqueued frames may still read them, so drain the stream first (a scene change, not a per-frame event)
    bool any_dead = false;
    for (char l : trio_live) any_dead |= !l;
    if (any_dead) {
        HIPOK(hipStreamSynchronize(ctx->stream));
        std::vector<pt_ctx::TrioRec> keep;
        for (size_t k = 0; k < ctx->trios.size(); k++) { if (trio_live[k]) keep.push_back(ctx->trios[k]); else hipFree((void*)ctx->trios[k].ptr); }
        ctx->trios.swap(keep);
    }
    return PT_OK;
}

// diagnostic (not part of include/mipt.h): how many materials of the current table read the interleaved footprint
extern "C" int pt_debug_interleaved_materials(const pt_ctx* ctx) {
    int n = 0;
    if (ctx) for (const RMat& r : ctx->rmats_host) n += (r.bound_mask & RM_TRIO) ? 1 : 0;
    return n;
}

extern "C" int pt_debug_interleaved_emissive(const pt_ctx* ctx) {          // ... and how many of them carry their emissive texture in it
    int n = 0;
    if (ctx) for (const RMat& r : ctx->rmats_host) n += (r.bound_mask & RM_TRIO_EMISSIVE) ? 1 : 0;
    return n;
}

// test hook (not part of include/mipt.h): what the product's traversal finds for caller-supplied rays (host arrays: 8 floats per ray in,
// 8 floats per ray out, pt_kernel.hip k_debug_intersect).  mode 0 = TraceRay's closest hit, 1 = TraceShadowRay's occlusion search.
extern "C" int pt_debug_intersect(pt_ctx* ctx, const float* rays, uint32_t n, uint32_t ray_flags, int mode, float* out) {
    if (!ctx || (n && (!rays || !out)) || mode < 0 || mode > 1) return PT_ERR_INVALID_ARGUMENT;
    ENTER(ctx);
    if (ctx->accel_state != ACCEL_CLEAN || !ctx->accel_built || ctx->instances_dirty) { int r = Pathtracer::BuildAccel(ctx); if (r) return r; }
    if (n == 0) return PT_OK;
    SceneRec sc;
    memset(&sc, 0, sizeof(sc));
    sc.rmats = ctx->d_rmats; sc.lights = ctx->d_lights; sc.instances = ctx->d_instances;
    sc.n_materials = (uint32_t)ctx->n_materials; sc.n_instances = (uint32_t)ctx->instances.size();
    sc.nodes = ctx->d_nodes; sc.tris = ctx->d_tris; sc.shade = ctx->d_shade; sc.root = ctx->root; sc.num_tris = ctx->n_tris;
    sc.sheen_e = ctx->d_sheen; sc.srgb_lut = ctx->d_srgb; sc.tangent_lut = ctx->d_tangent_lut;
    const uint32_t lanes = (n + 255u) & ~255u;
    float *d_rays = nullptr, *d_out = nullptr; int32_t* d_deep = nullptr;
    auto done = [&](int code, const std::string& why) { hipFree(d_rays); hipFree(d_out); hipFree(d_deep); return code == PT_OK ? PT_OK : ctx->fail(code, why); };
    if (hipMalloc((void**)&d_rays, (size_t)n * 32) != hipSuccess || hipMalloc((void**)&d_out, (size_t)n * 32) != hipSuccess) { (void)hipGetLastError(); return done(PT_ERR_OUT_OF_MEMORY, "pt_debug_intersect: ray buffers"); }
    if ((int)ctx->stack_need > traversal_stack_capacity()) {
        const uint32_t entries = (ctx->stack_need - (uint32_t)traversal_stack_capacity() + 7u) & ~7u;
        if (hipMalloc((void**)&d_deep, (size_t)entries * lanes * 4) != hipSuccess) { (void)hipGetLastError(); return done(PT_ERR_OUT_OF_MEMORY, "pt_debug_intersect: deep stack"); }
        sc.deep_stack = d_deep; sc.deep_entries = entries; sc.deep_lanes = lanes;
    }
    hipError_t e = hipMemcpyAsync(d_rays, rays, (size_t)n * 32, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) { launch_debug_intersect(sc, d_rays, n, ray_flags, mode, d_out, ctx->stream); e = hipGetLastError(); }
    if (e == hipSuccess && getenv("MIPT_DEBUG_INTERSECT_TIMING")) {           // probe (tools/ray_order_probe.py): the same launch timed, 5 repeats
        hipEvent_t ev[2]; hipEventCreate(&ev[0]); hipEventCreate(&ev[1]);
        hipEventRecord(ev[0], ctx->stream);
        for (int k = 0; k < 5; k++) launch_debug_intersect(sc, d_rays, n, ray_flags, mode, d_out, ctx->stream);
        hipEventRecord(ev[1], ctx->stream); hipEventSynchronize(ev[1]);
        float ms = 0; hipEventElapsedTime(&ms, ev[0], ev[1]);
        fprintf(stderr, "pt_debug_intersect: %u rays, %.3f ms per launch, %.1f Mrays/s\n", n, ms / 5, n / (ms / 5) * 1e-3);
        hipEventDestroy(ev[0]); hipEventDestroy(ev[1]);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, (size_t)n * 32, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    return done(e == hipSuccess ? PT_OK : PT_ERR_DEVICE, std::string("pt_debug_intersect: ") + hipGetErrorString(e));
}

int pt_scene_set_lights(pt_ctx* ctx, const pt_light* l, int count) {
    if (!ctx || (count > 0 && !l) || count < 0) return PT_ERR_INVALID_ARGUMENT;
    ENTER(ctx);
    std::vector<pt_light> h(l, l + count);
    HIPOK(upload_table(ctx, ctx->d_lights, ctx->lights_cap, h));
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
        if (!live_buffer(ctx, h)) return false;
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
    // What the new table asks of the acceleration structure (the reference rebuilds its TLAS every frame, Pathtracer.cpp:282, which
    // makes a moved instance free; here instances are flattened into one tree):
    //   identical table            -> nothing (gs_frame re-sends the table every frame, also for a static scene);
    //   same triangles, rows differ -> refit of the rows that differ (transform, flags, streams re-pointed at same-sized buffers);
    //   anything else              -> full build.
    const std::vector<InstanceRec>& old = ctx->instances;
    bool same_shape = ctx->accel_built && old.size() == recs.size() && ctx->n_tris == (uint32_t)tris;
    for (size_t i = 0; same_shape && i < recs.size(); i++)
        same_shape = old[i].tri_count == recs[i].tri_count && old[i].tri_offset == recs[i].tri_offset;
    if (same_shape) {
        if (ctx->touched.size() != recs.size()) ctx->touched.assign(recs.size(), 0);
        bool any = false;
        for (size_t i = 0; i < recs.size(); i++) {
            if (memcmp(&old[i], &recs[i], sizeof(InstanceRec)) == 0) continue;
            any = true;
            // a row that differs only in its material id leaves every packet as it is (the packets name the instance, not the material)
            InstanceRec a = old[i], b = recs[i];
            a.gpu.material_id = b.gpu.material_id = 0;
            if (memcmp(&a, &b, sizeof(InstanceRec)) != 0) ctx->touched[i] = 1;
        }
        if (!any) return PT_OK;
        bool refit = false;
        for (uint8_t t : ctx->touched) refit = refit || t != 0;
        if (refit && ctx->accel_state == ACCEL_CLEAN) ctx->accel_state = ACCEL_REFIT;
    } else {
        ctx->accel_state = ACCEL_REBUILD;
        ctx->touched.assign(recs.size(), 0);
    }
    ctx->instances.swap(recs);
    ctx->instances_dirty = true;
    ctx->n_tris = (uint32_t)tris;
    return PT_OK;
}

int pt_env_create(pt_ctx* ctx, const float* rgb, int width, int height, int* env_out) {
    if (!ctx || !rgb || !env_out || width <= 0 || height <= 0) return PT_ERR_INVALID_ARGUMENT;
    ENTER(ctx);
    float* d = nullptr;
    size_t bytes = (size_t)width * height * 12;
    HIPOK(hipMalloc((void**)&d, bytes));
    hipError_t e = hipMemcpy(d, rgb, bytes, hipMemcpyHostToDevice);
    EnvDevice* env = new EnvDevice();
    if (!e) e = env_build(*env, d, width, height, ctx->stream);
    if (!e) e = hipStreamSynchronize(ctx->stream);
    hipFree(d);
    if (e) { env_free(*env); delete env; return ctx->fail(PT_ERR_DEVICE, std::string("pt_env_create: ") + hipGetErrorString(e)); }
    if (!ctx->free_envs.empty()) { *env_out = ctx->free_envs.back(); ctx->free_envs.pop_back(); ctx->envs[*env_out] = env; }
    else { ctx->envs.push_back(env); *env_out = (int)ctx->envs.size() - 1; }
    return PT_OK;
}

static bool live_env(const pt_ctx* ctx, int env) { return env >= 0 && env < (int)ctx->envs.size() && ctx->envs[env] != nullptr; }

// EnvironmentMap::Destroy: the maps of one environment (the reference replaces its single environment map in place when a new
// image is loaded, Source/EnvironmentMap.cpp:84-130).
int pt_env_destroy(pt_ctx* ctx, int env) {
    if (!ctx) return PT_ERR_INVALID_ARGUMENT;
    if (!live_env(ctx, env)) return ctx->fail(PT_ERR_BAD_HANDLE, "pt_env_destroy");
    ENTER(ctx);
    HIPOK(hipStreamSynchronize(ctx->stream));
    env_free(*ctx->envs[env]);
    delete ctx->envs[env];
    ctx->envs[env] = nullptr;
    ctx->free_envs.push_back(env);
    return PT_OK;
}

int pt_env_read(pt_ctx* ctx, int env, int* cube_size_out, uint16_t* cube, float* pyramid) {
    if (!ctx) return PT_ERR_INVALID_ARGUMENT;
    if (!live_env(ctx, env)) return ctx->fail(PT_ERR_BAD_HANDLE, "pt_env_read");
    ENTER(ctx);
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
    ENTER(ctx);
    return Pathtracer::BuildAccel(ctx);
}

int pt_accel_request_rebuild(pt_ctx* ctx) {
    if (!ctx) return PT_ERR_INVALID_ARGUMENT;
    ctx->accel_state = ACCEL_REBUILD;
    return PT_OK;
}

int pt_set_accel_builder(pt_ctx* ctx, int builder) {
    if (!ctx || builder < PT_BUILDER_LBVH || builder > PT_BUILDER_PLOC_REINSERT) return PT_ERR_INVALID_ARGUMENT;
    if (ctx->scratch.builder != builder) { ctx->scratch.builder = builder; ctx->accel_state = ACCEL_REBUILD; }
    return PT_OK;
}

int pt_skin_run(pt_ctx* ctx, const pt_skin_params* params, const pt_bone* bones, int bone_count) {
    if (!ctx || !params) return PT_ERR_INVALID_ARGUMENT;
    ENTER(ctx);
    return GpuSkin::Run(ctx, params, bones, bone_count);
}

int pt_trace(pt_ctx* ctx, const pt_settings* settings, const pt_execute_params* params) {
    if (!ctx || !settings || !params) return PT_ERR_INVALID_ARGUMENT;
    if (params->width == 0 || params->height == 0) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "zero resolution");
    ENTER(ctx);
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

int pt_enable_stage_timing(pt_ctx* ctx, int enable) {
    if (!ctx) return PT_ERR_INVALID_ARGUMENT;
    ctx->stage_timing = enable != 0;
    if (!ctx->stage_timing) ctx->timers.used = 0;
    return PT_OK;
}

int pt_reset_stats(pt_ctx* ctx) {
    if (!ctx) return PT_ERR_INVALID_ARGUMENT;
    ENTER(ctx);
    HIPOK(hipMemsetAsync(ctx->d_counters, 0, sizeof(Counters), ctx->stream));
    return PT_OK;
}

int pt_get_stats(pt_ctx* ctx, pt_stats* out) {
    if (!ctx || !out) return PT_ERR_INVALID_ARGUMENT;
    ENTER(ctx);
    HIPOK(hipStreamSynchronize(ctx->stream));
    Counters c;
    HIPOK(hipMemcpy(&c, ctx->d_counters, sizeof(c), hipMemcpyDeviceToHost));
    memset(out, 0, sizeof(*out));
    out->rays_primary = c.rays_primary; out->rays_bounce = c.rays_bounce; out->rays_shadow = c.rays_shadow;
    out->rays = c.rays_primary + c.rays_bounce + c.rays_shadow;
    out->nodes_visited = c.nodes + c.nodes_shadow; out->tris_tested = c.tris + c.tris_shadow; out->closest_hits = c.hits; out->texture_taps = c.taps;
    out->nodes_visited_shadow = c.nodes_shadow; out->tris_tested_shadow = c.tris_shadow;
    if (ctx->have_trace) hipEventElapsedTime(&out->trace_ms, ctx->ev_trace[0], ctx->ev_trace[1]);
    if (ctx->have_accel) hipEventElapsedTime(&out->accel_ms, ctx->ev_accel[0], ctx->ev_accel[1]);
    if (ctx->have_skin) hipEventElapsedTime(&out->skin_ms, ctx->ev_skin[0], ctx->ev_skin[1]);
    for (size_t k = 0; k < ctx->timers.used; k++) {          // per-stage times of the last pt_trace (pt_enable_stage_timing)
        float ms = 0;
        if (hipEventElapsedTime(&ms, ctx->timers.ev[k], ctx->timers.ev[k + 1]) == hipSuccess) out->stage_ms[ctx->timers.kind[k]] += ms;
    }
    out->accumulated_frames = ctx->accumulated_frames;
    out->bvh_nodes = ctx->wide_nodes;
    out->bvh_triangles = ctx->n_tris;
    out->bvh_stack_need = ctx->stack_need;
    out->bvh_stack_capacity = (int)ctx->stack_need > traversal_stack_capacity() ? ((ctx->stack_need - (uint32_t)traversal_stack_capacity() + 7u) & ~7u) + (uint32_t)traversal_stack_capacity()
                                                                               : (uint32_t)traversal_stack_capacity();
    out->accel_builder_fallbacks = ctx->scratch.fallbacks;
    out->deep_stack_pushes = c.deep_pushes;
    out->accel_builds = ctx->accel_builds; out->accel_refits = ctx->accel_refits;
    if (c.stack_overflow) return ctx->fail(PT_ERR_CAPACITY, "traversal stack overflow: " + std::to_string(c.stack_overflow) + " pushes dropped");
    return PT_OK;
}

int pt_readback(pt_ctx* ctx, const void* device_rgba32f, uint32_t width, uint32_t height, float* host) {
    if (!ctx || !device_rgba32f || !host) return PT_ERR_INVALID_ARGUMENT;
    ENTER(ctx);
    HIPOK(hipStreamSynchronize(ctx->stream));
    HIPOK(hipMemcpy(host, device_rgba32f, (size_t)width * height * 16, hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_tonemap(pt_ctx* ctx, const pt_tonemap_config* cfg, const void* device_rgba32f, uint32_t width, uint32_t height, float* host_rgb, uint8_t* host_rgba8) {
    if (!ctx || !cfg || !device_rgba32f || width == 0 || height == 0) return PT_ERR_INVALID_ARGUMENT;
    ENTER(ctx);
    const size_t n = (size_t)width * height, need = n * 16;      // float RGB (12 B) + RGBA8 (4 B) per pixel, kept between calls
    if (need > ctx->tonemap_cap) {
        HIPOK(hipStreamSynchronize(ctx->stream));
        hipFree(ctx->d_tonemap); ctx->d_tonemap = nullptr; ctx->tonemap_cap = 0;
        HIPOK(hipMalloc(&ctx->d_tonemap, need));
        ctx->tonemap_cap = need;
    }
    float* d_rgb = host_rgb ? (float*)ctx->d_tonemap : nullptr;
    uint32_t* d_q = host_rgba8 ? (uint32_t*)((char*)ctx->d_tonemap + n * 12) : nullptr;
    launch_tonemap((const float4*)device_rgba32f, width, height, *cfg, d_rgb, d_q, ctx->stream);
    HIPOK(hipGetLastError());
    HIPOK(hipStreamSynchronize(ctx->stream));
    if (host_rgb) HIPOK(hipMemcpy(host_rgb, d_rgb, n * 12, hipMemcpyDeviceToHost));
    if (host_rgba8) HIPOK(hipMemcpy(host_rgba8, d_q, n * 4, hipMemcpyDeviceToHost));
    return PT_OK;
}

// ---- the per-frame exchange of the tile-sharded renderer (exchange.hip) ------------------------------------------------------
int pt_exchange_unique_id(void* id_out) {
    if (!id_out) return PT_ERR_INVALID_ARGUMENT;
    std::string err;
    return exchange_unique_id(id_out, err);
}

int pt_exchange_create(pt_ctx* ctx, int rank, int world, const void* unique_id) {
    if (!ctx || world < 1 || rank < 0 || rank >= world) return PT_ERR_INVALID_ARGUMENT;
    ENTER(ctx);
    if (ctx->exchange) { HIPOK(hipStreamSynchronize(ctx->stream)); exchange_free(ctx->exchange); ctx->exchange = nullptr; }
    std::string err;
    const int rc = exchange_create(&ctx->exchange, rank, world, unique_id, err);
    return rc == PT_OK ? PT_OK : ctx->fail(rc, err);
}

int pt_exchange_probe(void) {
    std::string err;
    return exchange_probe(err);
}

int pt_exchange_create_loopback(pt_ctx* ctx, int rank, int world, uint64_t group) {
    if (!ctx || world < 1 || rank < 0 || rank >= world) return PT_ERR_INVALID_ARGUMENT;
    ENTER(ctx);
    if (ctx->exchange) { HIPOK(hipStreamSynchronize(ctx->stream)); exchange_free(ctx->exchange); ctx->exchange = nullptr; }
    std::string err;
    const int rc = exchange_create_loopback(&ctx->exchange, rank, world, group, ctx->device, err);
    return rc == PT_OK ? PT_OK : ctx->fail(rc, err);
}

int pt_exchange_frame(pt_ctx* ctx, const void* local_image, void* frame, uint32_t width, uint32_t height, int mode, int dst_rank) {
    if (!ctx || !local_image || width == 0 || height == 0) return PT_ERR_INVALID_ARGUMENT;
    if (!ctx->exchange) return ctx->fail(PT_ERR_NOT_READY, "pt_exchange_frame: call pt_exchange_create first");
    if (mode != PT_EXCHANGE_GATHER && mode != PT_EXCHANGE_REDUCE) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "pt_exchange_frame: mode");
    ENTER(ctx);
    std::string err;
    const int rc = exchange_frame_checked(ctx, local_image, frame, width, height, mode, dst_rank, err);
    return rc == PT_OK ? PT_OK : ctx->fail(rc, err);
}

int pt_exchange_destroy(pt_ctx* ctx) {
    if (!ctx) return PT_ERR_INVALID_ARGUMENT;
    ENTER(ctx);
    if (ctx->exchange) { HIPOK(hipStreamSynchronize(ctx->stream)); exchange_free(ctx->exchange); ctx->exchange = nullptr; }
    return PT_OK;
}

size_t pt_tiles_packed_bytes(uint32_t width, uint32_t height, uint32_t tile_rank, uint32_t tile_rank_count) {
    if (tile_rank_count == 0 || tile_rank >= tile_rank_count) return 0;
    return (size_t)tiles_of_rank(width, height, tile_rank, tile_rank_count) * 256 * 16;
}

int pt_tiles_pack(pt_ctx* ctx, const void* image, uint32_t width, uint32_t height, uint32_t tile_rank, uint32_t tile_rank_count, void* packed) {
    if (!ctx || !image || !packed || width == 0 || height == 0 || tile_rank_count == 0 || tile_rank >= tile_rank_count) return PT_ERR_INVALID_ARGUMENT;
    ENTER(ctx);
    HIPOK(tiles_pack(image, width, height, tile_rank, tile_rank_count, packed, ctx->stream));
    return PT_OK;
}

int pt_tiles_unpack(pt_ctx* ctx, const void* packed, uint32_t width, uint32_t height, uint32_t tile_rank, uint32_t tile_rank_count, void* image) {
    if (!ctx || !image || !packed || width == 0 || height == 0 || tile_rank_count == 0 || tile_rank >= tile_rank_count) return PT_ERR_INVALID_ARGUMENT;
    ENTER(ctx);
    HIPOK(tiles_unpack(packed, width, height, tile_rank, tile_rank_count, image, ctx->stream));
    return PT_OK;
}

}  // extern "C"
