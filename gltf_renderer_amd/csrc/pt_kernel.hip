// pt_kernel.hip -- the path-tracing kernels for gfx950 (MI355X).
//
// What the reference does with five DXR shaders called recursively (RayGeneration -> TraceRay ->
// ClosestHit / AnyHit / Miss / ShadowAnyHit / ShadowMiss, Source/Shaders/PathTracer.lib.hlsl:744-1085)
// is done here by pure-compute kernels in two interchangeable arrangements of the same per-vertex code
// (pt_vertex.h, pt_traverse.h):
//
//  * WAVEFRONT (default): the frame advances one path vertex at a time through three stages --
//    trace (closest hit) -> shade -> trace (shadow) -- over SoA ray / hit / path-state arrays in HBM.
//    Surviving paths are compacted into the next queue with a wave64 ballot + one atomic per wave, so every
//    stage runs with full waves; the trace stages are small-register kernels that run at high occupancy to
//    hide the dependent-load latency of BVH traversal, the heavy material code runs only on lanes with a hit.
//    MI355X has the HBM bandwidth to spare (the state traffic is ~300 B per vertex against 8 TB/s); it does
//    not have RT cores, so occupancy and wave coherence are what buy ray throughput.
//  * MEGAKERNEL: one lane owns one pixel-sample for its whole life (state machine with ONE traversal site).
//    Kept as the simplest correct arrangement and as the A/B baseline.
//
// Work mapping (both): a rank renders the 16x16 tiles t with t % tile_rank_count == tile_rank; slot s of a
// rank is pixel (s & 255) of its (s >> 8)-th tile, one wave64 per 8x8 quadrant (coherent primary rays).
#include "pt_vertex.h"
#include "pt_host.h"

namespace pt {

enum { KIND_CLOSEST = 0, KIND_SHADOW_ENV = 1, KIND_SHADOW_LIGHT = 2 };

// =================================================================================================
// MEGAKERNEL
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void pt_megakernel(SceneRec sc, FrameConstants fc, float4* __restrict__ output, Counters* __restrict__ counters) {
    __shared__ int s_stack[kStackLds * kBlock];
    int* my_stack = s_stack + threadIdx.x;
    uint32_t px, py;
    const bool in_image = slot_pixel(fc, blockIdx.x * kBlock + threadIdx.x, px, py);
    bool alive = in_image;
    const uint32_t flags = fc.flags;
    LaneStats st = {0, 0, 0, 0};
    unsigned n_primary = 0, n_bounce = 0, n_shadow = 0, n_hits = 0;
    PathState ps;
    ps.beta = v3(1); ps.thr = v3(1); ps.prev_pdf = 0; ps.rc = 0; ps.bounce = 0; ps.prev_mis = false;
    vec3 L = v3(0);
    Ray ray;
    ray.o = v3(0); ray.d = v3(0, 0, 1); ray.tmin = 0; ray.tmax = 0;
    uint32_t rf = 0, rmask = 0xff;
    int kind = KIND_CLOSEST;
    Followups fu;
    fu.q_env = fu.q_light = fu.q_bounce = false;
    fu.pend_env = fu.pend_light = v3(0);
    if (alive) {
        ray = camera_ray(fc, fc.seed, px, py, ps.rc);
        rf = (flags & PT_FLAG_CULL_BACKFACE) ? RF_CULL_BACK : 0;
        n_primary++;
    }
    while (alive) {
        HitRec hit;
        float transmission = 0.0f;
        bool got;
        if (kind == KIND_CLOSEST) {
            got = traverse<COUNT>(sc, my_stack, ray, rf, rmask, 0, hit, transmission, st);
        } else {
            bool alpha_shadow = (kind == KIND_SHADOW_LIGHT) && (flags & PT_FLAG_ALPHA_SHADOWS);      // TraceShadowRay :724-742
            uint32_t srf = (flags & PT_FLAG_CULL_BACKFACE) ? RF_CULL_BACK : 0;
            if (alpha_shadow) { transmission = 1.0f; srf |= RF_FORCE_NON_OPAQUE; }
            else srf |= RF_ACCEPT_FIRST;
            got = traverse<COUNT>(sc, my_stack, ray, srf, 0xff, 1, hit, transmission, st);
            if (!got) transmission = 1.0f;                                                           // ShadowMiss :1081-1085
        }
        // The reference multiplies the light colour by the shadow transmission BEFORE `if (any(color > 0))` and never
        // evaluates the BSDF of an occluded sample (:933-935, :949-951): an occluded sample must contribute nothing even
        // when its (pre-evaluated) pending term is NaN, hence the guard instead of a bare multiply by 0.
        if (kind == KIND_SHADOW_ENV) { if (transmission > 0.0f) L += fu.pend_env * transmission; }
        else if (kind == KIND_SHADOW_LIGHT) { if (transmission > 0.0f) L += fu.pend_light * transmission; }
        else if (!got) { L += shade_miss(sc, fc, ray.d, ps); alive = false; }
        else {
            n_hits++;
            bool done = shade_closest_hit(sc, fc, fc.seed, px, py, ray, hit, load_shade_packet_raw(sc.shade + hit.tri), sc.shade + hit.tri, ps, fu, st.taps);
            if (fu.overwrite) L = v3(0);
            L += fu.add;
            n_shadow += fu.counted_shadow;
            if (done) alive = false;
        }
        if (alive) {
            if (fu.q_env) {
                fu.q_env = false; kind = KIND_SHADOW_ENV;
                ray.o = fu.origin_above; ray.tmin = 0; ray.d = fu.env_dir; ray.tmax = fc.max_ray_length;
                n_shadow++;
            } else if (fu.q_light) {
                fu.q_light = false; kind = KIND_SHADOW_LIGHT;
                ray.o = fu.origin_above; ray.tmin = 0; ray.d = fu.light_dir; ray.tmax = fc.max_ray_length;
                n_shadow++;
            } else if (fu.q_bounce) {                                                                // TraceBounceRay :669-678
                fu.q_bounce = false; kind = KIND_CLOSEST;
                ray.o = fu.b_o; ray.tmin = 0; ray.d = fu.b_d; ray.tmax = fc.max_ray_length;
                ps.beta = fu.b_beta; ps.thr = fu.b_thr; ps.prev_pdf = fu.b_pdf; ps.prev_mis = fu.b_mis;
                ps.bounce++;
                rmask = (flags & PT_FLAG_INDIRECT_ENVIRONMENT_ONLY) ? 0 : 0xff;
                rf = (flags & PT_FLAG_CULL_BACKFACE) ? RF_CULL_FRONT : 0;                             // (sic) quirk q2
                n_bounce++;
            } else alive = false;
        }
    }
    if (in_image) write_pixel(fc, fc.accumulated_frames, output, px, py, L);
    flush_counters(counters, threadIdx.x & 63, n_primary, n_bounce, n_shadow, n_hits, st);
    if (st.deep) atomicAdd(&counters->deep_pushes, (unsigned long long)st.deep);
}

template __global__ void pt_megakernel<false>(SceneRec, FrameConstants, float4*, Counters*);
template __global__ void pt_megakernel<true>(SceneRec, FrameConstants, float4*, Counters*);

void launch_megakernel(const SceneRec& sc, const FrameConstants& fc, float4* output, Counters* counters, bool count, hipStream_t stream) {
    if (fc.my_tiles == 0) return;
    dim3 grid(fc.my_tiles), block(kBlock);
    if (count) hipLaunchKernelGGL(pt_megakernel<true>, grid, block, 0, stream, sc, fc, output, counters);
    else hipLaunchKernelGGL(pt_megakernel<false>, grid, block, 0, stream, sc, fc, output, counters);
}

// Test hook (pt_debug_intersect): the product's traversal on caller-supplied rays, one ray per lane -- what TraceRay / TraceShadowRay find,
// without the shading around them.  rays: 8 floats each (origin, tmin, direction, tmax); out: 8 floats each (committed, t, u, v, instance,
// primitive, front face, transmission).
__global__ __launch_bounds__(kBlock) void k_debug_intersect(SceneRec sc, const float* __restrict__ rays, uint32_t n, uint32_t rf, int mode, float* __restrict__ out) {
    __shared__ int s_stack[kStackLds * kBlock];
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float* q = rays + (size_t)i * 8;
    Ray r; r.o = v3(q[0], q[1], q[2]); r.tmin = q[3]; r.d = v3(q[4], q[5], q[6]); r.tmax = q[7];
    HitRec hit; LaneStats st = {0, 0, 0, 0, 0};
    float transmission = (mode == 1 && (rf & RF_FORCE_NON_OPAQUE)) ? 1.0f : 0.0f;
    const bool got = traverse<false>(sc, s_stack + threadIdx.x, r, rf, 0xff, mode, hit, transmission, st);
    float* o = out + (size_t)i * 8;
    const bool have = got && hit.tri >= 0;
    o[0] = got ? 1.0f : 0.0f; o[1] = have ? hit.t : 0.0f; o[2] = have ? hit.u : 0.0f; o[3] = have ? hit.v : 0.0f;
    o[4] = have ? (float)sc.tris[hit.tri].inst : -1.0f; o[5] = have ? (float)sc.tris[hit.tri].prim : -1.0f; o[6] = (have && hit.front) ? 1.0f : 0.0f;
    o[7] = transmission;
}
void launch_debug_intersect(const SceneRec& sc, const float* d_rays, uint32_t n, uint32_t rf, int mode, float* d_out, hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_debug_intersect, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, sc, d_rays, n, rf, mode, d_out);
}

// Test hook (pt_debug_math): the kernels' own math routines on caller-supplied arguments, for the bit-for-bit comparison with the oracle's.
// op: 0 atan2(a, b)  1 pow(a, b)  2 exp(a)  3 log2(a)  4 exp2(a)  5 sin(a)  6 cos(a)  7 a / b by fdiv  8 pow5(a)
__global__ __launch_bounds__(256) void k_debug_math(int op, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float x = a[i], y = b[i];
    float r = 0, s, c;
    switch (op) {
        case 0: r = pt_atan2(x, y); break;
        case 1: r = hpow(x, y); break;
        case 2: r = pt_exp(x); break;
        case 3: r = co_log2(x); break;
        case 4: r = co_exp2(x); break;
        case 5: pt_sincos(x, s, c); r = s; break;
        case 6: pt_sincos(x, s, c); r = c; break;
        case 7: r = fdiv(x, y); break;
        case 8: r = hpow5(x); break;
    }
    out[i] = r;
}
}  // namespace pt
extern "C" int pt_debug_math(int op, const float* a, const float* b, float* out, uint32_t n) {
    if (n == 0) return 0;
    float *da = nullptr, *db = nullptr, *dc = nullptr;
    int rc = -3;
    if (hipMalloc(&da, (size_t)n * 4) == hipSuccess && hipMalloc(&db, (size_t)n * 4) == hipSuccess && hipMalloc(&dc, (size_t)n * 4) == hipSuccess &&
        hipMemcpy(da, a, (size_t)n * 4, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(db, b, (size_t)n * 4, hipMemcpyHostToDevice) == hipSuccess) {
        hipLaunchKernelGGL(pt::k_debug_math, dim3((n + 255) / 256), dim3(256), 0, nullptr, op, (const float*)da, (const float*)db, dc, n);
        if (hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess && hipMemcpy(out, dc, (size_t)n * 4, hipMemcpyDeviceToHost) == hipSuccess) rc = 0;
    }
    hipFree(da); hipFree(db); hipFree(dc);
    return rc;
}
namespace pt {

// (sin, cos) table of the packed tangent angle, filled by the decoder's own expression (pt_shading.h)
__global__ __launch_bounds__(256) void k_tangent_lut(float2* __restrict__ out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < 1024u) out[k] = tangent_sincos_compute(k);
}
hipError_t build_tangent_lut(float2* d_lut, hipStream_t stream) {
    hipLaunchKernelGGL(k_tangent_lut, dim3(4), dim3(256), 0, stream, d_lut);
    return hipGetLastError();
}

}  // namespace pt
