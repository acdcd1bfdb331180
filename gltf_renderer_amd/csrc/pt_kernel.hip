// pt_kernel.hip -- the path-tracing megakernel for gfx950 (MI355X).
//
// What the reference does with five DXR shaders called recursively (RayGeneration -> TraceRay ->
// ClosestHit / AnyHit / Miss / ShadowAnyHit / ShadowMiss, Source/Shaders/PathTracer.lib.hlsl:744-1085)
// is done here by one pure-compute kernel: every lane owns one pixel-sample and runs an iterative state
// machine with ONE traversal site.  At a closest hit all random numbers of that vertex are drawn in the
// reference's order (env NEE, light NEE, BSDF, Russian roulette) and the up-to-three follow-up rays
// (environment shadow, light shadow, bounce) are queued in registers, each shadow ray with its
// pre-multiplied pending contribution; the recursion's `color += weight * child` becomes
// `L += beta * contribution` with beta the running product of bounce weights.
//
// Work mapping: one 256-thread workgroup per 16x16 pixel tile, one wave64 per 8x8 quadrant (coherent
// primary rays, 2-D locality in textures and BVH); tile t belongs to rank (t % tile_rank_count) so N GPUs
// shard a frame with no data-path exchange.
#include "pt_traverse.h"
#include "pt_host.h"

namespace pt {

enum { KIND_CLOSEST = 0, KIND_SHADOW_ENV = 1, KIND_SHADOW_LIGHT = 2 };

template <bool COUNT>
__global__ __launch_bounds__(kBlock) void pt_megakernel(SceneRec sc, FrameConstants fc, float4* __restrict__ output, Counters* __restrict__ counters) {
    __shared__ int s_stack[kStackLds * kBlock];
    int* my_stack = s_stack + threadIdx.x;

    // tile -> pixel
    const uint32_t tile = fc.tile_rank + blockIdx.x * fc.tile_rank_count;
    const uint32_t tx = tile % fc.tiles_x, ty = tile / fc.tiles_x;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t px = tx * PT_TILE + (wave & 1) * 8 + (lane & 7);
    const uint32_t py = ty * PT_TILE + (wave >> 1) * 8 + (lane >> 3);
    bool alive = (px < fc.res_x) && (py < fc.res_y) && (tile < fc.tiles_x * fc.tiles_y);
    const bool in_image = alive;
    const uint32_t flags = fc.flags;

    LaneStats st = {0, 0, 0, 0};
    unsigned n_primary = 0, n_bounce = 0, n_shadow = 0, n_hits = 0;

    // ---- path state (Payload, PathTracer.lib.hlsl:110-117, made iterative)
    int rc = 0, bounce = 0;
    vec3 L = v3(0), beta = v3(1), thr = v3(1);
    float prev_pdf = 0;
    bool prev_mis = false;
    // ---- current ray
    Ray ray;
    uint32_t rf = 0, rmask = 0xff;
    int kind = KIND_CLOSEST;
    // ---- queued follow-ups of the last closest hit
    vec3 pend_env = v3(0), pend_env_dir = v3(0, 0, 1), pend_light = v3(0), pend_light_dir = v3(0, 0, 1), origin_above = v3(0);
    bool q_env = false, q_light = false, q_bounce = false;
    vec3 b_o = v3(0), b_d = v3(0, 0, 1), b_beta = v3(0), b_thr = v3(0);
    float b_pdf = 0;
    bool b_mis = false;

    if (alive) {                                            // RayGeneration :744-758
        vec4 r = next_random(px, py, fc.seed, rc);
        float jx = r.x - 0.5f, jy = r.y - 0.5f;
        float cx = (((float)px + 0.5f + jx) / (float)fc.res_x) * 2 - 1;          // GenerateCameraRay :131-142
        float cy = (((float)py + 0.5f + jy) / (float)fc.res_y) * 2 - 1;
        cy = -cy;
        vec4 s = mul4(fc.clip_to_world, vec4{cx, cy, 1, 1});
        vec4 e = mul4(fc.clip_to_world, vec4{cx, cy, 0, 1});
        vec3 o = xyz(s) / s.w;
        vec3 d = xyz(e) / e.w - o;
        ray.o = o; ray.tmin = 0; ray.d = normalize(d); ray.tmax = length(d);
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
            bool alpha_shadow = (kind == KIND_SHADOW_LIGHT) && (flags & PT_FLAG_ALPHA_SHADOWS);     // TraceShadowRay :724-742
            uint32_t srf = (flags & PT_FLAG_CULL_BACKFACE) ? RF_CULL_BACK : 0;
            if (alpha_shadow) { transmission = 1.0f; srf |= RF_FORCE_NON_OPAQUE; }
            else srf |= RF_ACCEPT_FIRST;
            got = traverse<COUNT>(sc, my_stack, ray, srf, 0xff, 1, hit, transmission, st);
            if (!got) transmission = 1.0f;                                                          // ShadowMiss :1081-1085
        }

        // The reference multiplies the light colour by the shadow transmission BEFORE `if (any(color > 0))` and never
        // evaluates the BSDF of an occluded sample (:933-935, :949-951): an occluded sample must contribute nothing even
        // when its (pre-evaluated) pending term is NaN, hence the guard instead of a bare multiply by 0.
        if (kind == KIND_SHADOW_ENV) { if (transmission > 0.0f) L += pend_env * transmission; }
        else if (kind == KIND_SHADOW_LIGHT) { if (transmission > 0.0f) L += pend_light * transmission; }
        else if (!got) {
            // Miss :1037-1051
            vec3 c;
            if (flags & PT_FLAG_ENVIRONMENT_MAP) {
                c = sc.has_env ? fc.environment_intensity * sample_cube(sc.env.cube, sc.env.cube_n, ray.d) : v3(0);
                if ((flags & PT_FLAG_ENVIRONMENT_MIS) && prev_mis) {
                    float env_pdf = sc.has_env ? importance_map_pdf(sc.env, square_to_uv(sphere_to_square(normalize(ray.d)))) / (4 * kPi) : 0.f;
                    c *= prev_pdf / (prev_pdf + env_pdf);                                            // BalanceHeuristic :383-386
                }
            } else c = fc.environment_intensity * v3p(fc.environment_color);
            L += beta * c;
            alive = false;
        } else {
            // ---------------------------------------------------------------- ClosestHit :788-1007
            n_hits++;
            q_env = q_light = q_bounce = false;
            const float4* tp = (const float4*)sc.tris + (size_t)hit.tri * 3;
            const uint32_t inst_id = __float_as_uint(tp[0].w), prim = __float_as_uint(tp[1].w);
            const pt_mesh_instance& inst = sc.instances[inst_id].gpu;
            const pt_material& mat = sc.materials[inst.material_id];
            HitGeom va = get_vertex_attributes(sc, inst, prim, v3(1 - hit.u - hit.v, hit.u, hit.v));
            const int dbg = fc.debug_output;
            bool done = false;
            vec3 dbg_color = v3(0);
            if (dbg >= PT_DEBUG_OUTPUT_HIT_KIND && dbg <= PT_DEBUG_OUTPUT_TEXCOORD_1) {           // :806-840
                done = true;
                switch (dbg) {
                    case PT_DEBUG_OUTPUT_HIT_KIND: dbg_color = hit.front ? v3(1, 0, 0) : v3(0, 1, 0); break;
                    case PT_DEBUG_OUTPUT_VERTEX_COLOR: dbg_color = xyz(va.color); break;
                    case PT_DEBUG_OUTPUT_VERTEX_ALPHA: dbg_color = v3(va.color.w); break;
                    case PT_DEBUG_OUTPUT_VERTEX_NORMAL: dbg_color = (va.n + 1) / 2; break;
                    case PT_DEBUG_OUTPUT_VERTEX_TANGENT: dbg_color = (va.t + 1) / 2; break;
                    case PT_DEBUG_OUTPUT_VERTEX_BITANGENT: dbg_color = (va.bt + 1) / 2; break;
                    case PT_DEBUG_OUTPUT_TEXCOORD_0: dbg_color = v3(va.tc[0].x, va.tc[0].y, 0); break;
                    default: dbg_color = v3(va.tc[1].x, va.tc[1].y, 0); break;
                }
            }
            if (!done) {
                if (!hit.front) { va.ng = -va.ng; va.n = -va.n; va.t = -va.t; va.tw = -va.tw; }     // :842-846
                const vec3 intersection = ray.o + (ray.d * hit.t);                                  // :849
                const vec3 o_above = offset_ray(va.position, va.ng), o_below = offset_ray(va.position, -va.ng);
                const vec3 view = -normalize(ray.d);
                Surface sp = get_surface(sc, flags, mat, va, view, st.taps);
                if (dbg >= PT_DEBUG_OUTPUT_COLOR && dbg <= PT_DEBUG_OUTPUT_TRANSMISSIVE) {          // :863-917
                    done = true;
                    switch (dbg) {
                        case PT_DEBUG_OUTPUT_COLOR: dbg_color = sp.albedo; break;
                        case PT_DEBUG_OUTPUT_ALPHA: dbg_color = v3(sp.alpha); break;
                        case PT_DEBUG_OUTPUT_SHADING_NORMAL: dbg_color = (sp.n + 1) / 2; break;
                        case PT_DEBUG_OUTPUT_SHADING_TANGENT: dbg_color = (sp.at + 1) / 2; break;
                        case PT_DEBUG_OUTPUT_SHADING_BITANGENT: dbg_color = (sp.ab + 1) / 2; break;
                        case PT_DEBUG_OUTPUT_METALNESS: dbg_color = v3(sp.metalness); break;
                        case PT_DEBUG_OUTPUT_ROUGHNESS: dbg_color = v3(sqrtf(sp.ay)); break;
                        case PT_DEBUG_OUTPUT_SPECULAR: dbg_color = v3(sp.spec_factor); break;
                        case PT_DEBUG_OUTPUT_SPECULAR_COLOR: dbg_color = sp.spec_color; break;
                        case PT_DEBUG_OUTPUT_CLEARCOAT: dbg_color = v3(sp.clearcoat); break;
                        case PT_DEBUG_OUTPUT_CLEARCOAT_ROUGHNESS: dbg_color = v3(sp.cc_rough); break;
                        case PT_DEBUG_OUTPUT_CLEARCOAT_NORMAL: dbg_color = (sp.cc_n + 1) / 2; break;
                        default: dbg_color = v3(sp.transmissive); break;
                    }
                } else if (dbg == PT_DEBUG_OUTPUT_HEMISPHERE_VIEW_SIDE) {                           // :919-922
                    done = true;
                    dbg_color = dot(view, sp.n) > 0 ? v3(0, 1, 0) : v3(1, 0, 0);
                }
                if (!done) {
                    const Lobes lobes = lobe_probabilities(sp, view);
                    vec3 c = emissive_of(sc, mat, va.tc, st.taps);                                  // :925-926
                    origin_above = o_above;
                    // environment NEE :929-942 (SampleEnvironmentMap :688-703)
                    if (bounce < fc.max_bounces && (flags & PT_FLAG_ENVIRONMENT_MAP) && (flags & PT_FLAG_ENVIRONMENT_MIS)) {
                        vec4 r = next_random(px, py, fc.seed, rc);
                        float light_pdf = 1;
                        vec3 ldir = v3(0, 0, 1), lcol = v3(0);
                        if (sc.has_env) {
                            vec2 uv = sample_importance_map(sc.env, r.x, r.y, light_pdf);
                            ldir = square_to_sphere(uv_to_square(uv));
                            light_pdf /= 4 * kPi;
                            lcol = fc.environment_intensity * sample_cube(sc.env.cube, sc.env.cube_n, ldir);
                        }
                        vec3 contrib = v3(0);
                        if (any_gt0(lcol)) {
                            float bp = 0;
                            vec3 f = evaluate_bsdf(flags, sc.sheen_e, sp, lobes, va.ng, view, ldir, bp);
                            float mis = light_pdf / (light_pdf + bp);
                            contrib = (mis * f * lcol) / light_pdf;
                        }
                        if (flags & PT_FLAG_INDIRECT_ENVIRONMENT_ONLY) c += contrib;                // TraceShadowRay returns 1 untraced
                        else { q_env = true; pend_env = beta * contrib; pend_env_dir = ldir; }
                    }
                    // punctual-light NEE :945-956 (SamplePointLight :680-686)
                    if ((flags & PT_FLAG_POINT_LIGHTS) && fc.num_of_lights > 0) {
                        float u = next_random(px, py, fc.seed, rc).x;
                        uint32_t li = f2u(u * (float)fc.num_of_lights);
                        li = min(li, (uint32_t)(fc.num_of_lights - 1));                             // u may be exactly 1 (quirk q17)
                        float pdf = 1.0f / (float)fc.num_of_lights;
                        vec3 ldir, lcol;
                        light_ray(sc.lights[li], intersection, ldir, lcol);
                        vec3 contrib = v3(0);
                        if (any_gt0(lcol)) {
                            float bp = 0;
                            vec3 f = evaluate_bsdf(flags, sc.sheen_e, sp, lobes, va.ng, view, ldir, bp);
                            contrib = (lcol * f) / pdf;
                        }
                        if ((flags & PT_FLAG_SHADOW_RAYS) && !(flags & PT_FLAG_INDIRECT_ENVIRONMENT_ONLY)) { q_light = true; pend_light = beta * contrib; pend_light_dir = ldir; }
                        else c += contrib;
                    }
                    L += beta * c;
                    // BSDF sampling + Russian roulette :958-1006
                    if (bounce < fc.max_bounces) {
                        vec4 r = next_random(px, py, fc.seed, rc);
                        bool is_tr = false, use_mis = false;
                        float bp = 1;
                        vec3 l = v3(0);
                        vec3 f = sample_bsdf(flags, sc.sheen_e, sp, lobes, v3(r.x, r.y, r.z), view, l, bp, is_tr, use_mis);
                        vec3 weight = bp != 0 ? f / bp : v3(0);
                        vec3 throughput = thr * weight;
                        if (dbg >= PT_DEBUG_OUTPUT_BOUNCE_DIRECTION && dbg <= PT_DEBUG_BOUNCE_IS_TRANSMISSION) {   // :969-990
                            // payload.color is OVERWRITTEN here: emissive / NEE added above are discarded, no shadow rays count
                            switch (dbg) {
                                case PT_DEBUG_OUTPUT_BOUNCE_DIRECTION: dbg_color = 0.5f * (l + 1); break;
                                case PT_DEBUG_OUTPUT_BOUNCE_BSDF: dbg_color = f; break;
                                case PT_DEBUG_OUTPUT_BOUNCE_PDF: dbg_color = v3(bp); break;
                                case PT_DEBUG_OUTPUT_BOUNCE_WEIGHT: dbg_color = weight; break;
                                default: dbg_color = is_tr ? v3(0, 1, 0) : v3(1, 0, 0); break;
                            }
                            done = true;
                        } else if (any_gt0(throughput)) {
                            float ur = next_random(px, py, fc.seed, rc).x;                          // drawn even below min_bounces (quirk q6)
                            bool cont = bounce < fc.min_bounces;
                            if (!cont) {                                                            // RussianRoulette :712-722
                                float p = clampf(max3(throughput), fc.min_rr, fc.max_rr);
                                if (ur < p) { weight = weight / p; cont = true; }
                            }
                            if (cont) {
                                q_bounce = true;
                                b_o = is_tr ? o_below : o_above;
                                b_d = l;
                                b_beta = beta * weight;
                                b_thr = throughput * weight;                                        // weight applied twice (quirk q5)
                                b_pdf = bp; b_mis = use_mis;
                            }
                        }
                    }
                }
            }
            if (done) {
                // In the reference the debug cases of :969-990 run AFTER the NEE shadow rays were traced; their
                // results are overwritten, so only the ray counters differ: trace-count parity is kept by
                // counting them without tracing.
                if (dbg >= PT_DEBUG_OUTPUT_BOUNCE_DIRECTION && dbg <= PT_DEBUG_BOUNCE_IS_TRANSMISSION) {
                    n_shadow += (q_env ? 1 : 0) + (q_light ? 1 : 0);
                    L = v3(0);
                }
                L += beta * dbg_color;
                q_env = q_light = q_bounce = false;
                alive = false;
            }
        }

        // ---- next ray from the queue
        if (alive) {
            if (q_env) {
                q_env = false; kind = KIND_SHADOW_ENV;
                ray.o = origin_above; ray.tmin = 0; ray.d = pend_env_dir; ray.tmax = fc.max_ray_length;
                n_shadow++;
            } else if (q_light) {
                q_light = false; kind = KIND_SHADOW_LIGHT;
                ray.o = origin_above; ray.tmin = 0; ray.d = pend_light_dir; ray.tmax = fc.max_ray_length;
                n_shadow++;
            } else if (q_bounce) {                                                                  // TraceBounceRay :669-678
                q_bounce = false; kind = KIND_CLOSEST;
                ray.o = b_o; ray.tmin = 0; ray.d = b_d; ray.tmax = fc.max_ray_length;
                beta = b_beta; thr = b_thr; prev_pdf = b_pdf; prev_mis = b_mis;
                bounce++;
                rmask = (flags & PT_FLAG_INDIRECT_ENVIRONMENT_ONLY) ? 0 : 0xff;
                rf = (flags & PT_FLAG_CULL_BACKFACE) ? RF_CULL_FRONT : 0;                            // (sic) quirk q2
                n_bounce++;
            } else alive = false;
        }
    }

    // ---- RayGeneration epilogue :760-785
    if (in_image) {
        if (any_nan(L)) L = (flags & PT_FLAG_SHOW_NAN) ? v3(1, 0, 0) : v3(0);
        if (any_inf(L)) L = (flags & PT_FLAG_SHOW_INF) ? v3(1, 0, 0) : v3(0);
        if (flags & PT_FLAG_LUMINANCE_CLAMP) {
            float lum = luminance(L);
            if (lum > fc.luminance_clamp) L *= fc.luminance_clamp / lum;
        }
        float4* outp = output + ((size_t)py * fc.res_x + px);
        if ((flags & PT_FLAG_ACCUMULATE) && fc.accumulated_frames != 0) {
            float4 h = *outp;
            float blend = 1.0f / ((float)fc.accumulated_frames + 1.0f);
            *outp = make_float4(h.x + blend * (L.x - h.x), h.y + blend * (L.y - h.y), h.z + blend * (L.z - h.z), h.w + blend * (1.0f - h.w));
        } else *outp = make_float4(L.x, L.y, L.z, 1.0f);
    }

    // ---- counters: wave reduction, one atomic per wave per counter
    unsigned vals[8] = {n_primary, n_bounce, n_shadow, st.nodes, st.tris, n_hits, st.taps, st.overflow};
#pragma unroll
    for (int k = 0; k < 8; k++) {
        unsigned v = vals[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0 && v) atomicAdd(((unsigned long long*)counters) + k, (unsigned long long)v);
    }
}

template __global__ void pt_megakernel<false>(SceneRec, FrameConstants, float4*, Counters*);
template __global__ void pt_megakernel<true>(SceneRec, FrameConstants, float4*, Counters*);

void launch_megakernel(const SceneRec& sc, const FrameConstants& fc, float4* output, Counters* counters, bool count, hipStream_t stream) {
    if (fc.my_tiles == 0) return;
    dim3 grid(fc.my_tiles), block(kBlock);
    if (count) hipLaunchKernelGGL(pt_megakernel<true>, grid, block, 0, stream, sc, fc, output, counters);
    else hipLaunchKernelGGL(pt_megakernel<false>, grid, block, 0, stream, sc, fc, output, counters);
}

}  // namespace pt
