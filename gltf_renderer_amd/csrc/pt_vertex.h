// pt_vertex.h -- one path vertex of the iterative path tracer (shared by the megakernel and the wavefront
// shade stage).
//
// shade_closest_hit() is ClosestHit (Source/Shaders/PathTracer.lib.hlsl:788-1007) with the recursion removed:
// all random numbers of the vertex are drawn in the reference's order (env NEE, light NEE, BSDF, Russian
// roulette) and the up-to-three follow-up rays are returned as a queue, each shadow ray with its pending
// contribution already multiplied by the path weight beta.  shade_miss() is Miss (:1037-1051).
#pragma once
#include "pt_traverse.h"

namespace pt {

struct PathState {          // Payload (PathTracer.lib.hlsl:110-117) made iterative
    vec3 beta, thr;         // product of bounce weights / the reference's payload.throughput (feeds RR only)
    float prev_pdf;
    int rc, bounce;
    bool prev_mis;
};

struct Followups {
    vec3 add;               // beta-weighted radiance to add to L right away (emissive, untraced NEE, debug colour)
    bool overwrite;         // debug outputs of :969-990 overwrite payload.color
    bool q_env, q_light, q_bounce;
    vec3 pend_env, env_dir, pend_light, light_dir, origin_above;
    vec3 b_o, b_d, b_beta, b_thr;
    float b_pdf;
    bool b_mis;
    unsigned counted_shadow;   // shadow rays the reference traces but whose result a debug mode discards
    uint32_t light_index;      // the punctual light the light-NEE sample picked (occluder cache key)
    uint32_t hint_env, hint_light;   // occluder-cache words of the two shadow rays (fetched here, early, so that the push does not wait for them)
};

PT_DEV vec3 shade_miss(const SceneRec& sc, const FrameConstants& fc, vec3 dir, const PathState& ps) {
    const uint32_t flags = fc.flags;
    vec3 c;
    if (flags & PT_FLAG_ENVIRONMENT_MAP) {
        c = sc.has_env ? fc.environment_intensity * sample_cube(sc.env.cube, sc.env.cube_n, dir) : v3(0);
        if ((flags & PT_FLAG_ENVIRONMENT_MIS) && ps.prev_mis) {
            float env_pdf = sc.has_env ? fdiv(importance_map_pdf(sc.env, square_to_uv(sphere_to_square(normalize(dir)))), 4 * kPi) : 0.f;
            c *= fdiv(ps.prev_pdf, ps.prev_pdf + env_pdf);                                              // BalanceHeuristic :383-386
        }
    } else c = fc.environment_intensity * v3p(fc.environment_color);
    return ps.beta * c;
}

// Returns true when the path ends at this vertex for a debug output (fu.add holds beta * debug colour).
// -DPT_TIMING (pt_wavefront.hip only; tools/shade_sections.py): cycles per section of this function, summed per wave.
#ifdef PT_TIMING
extern __device__ unsigned long long pt_timing[12];
#define PT_TICK(K) { const unsigned long long _now = __builtin_readcyclecounter(); _sec[K] += _now - _t; _t = _now; }
#define PT_TICK_FLUSH() { if (__lane_id() == (unsigned)(__ffsll((long long)__ballot(1)) - 1)) { for (int _k = 0; _k < 11; _k++) atomicAdd(&pt_timing[_k], _sec[_k]); atomicAdd(&pt_timing[11], 1ull); } }
#else
#define PT_TICK(K)
#define PT_TICK_FLUSH()
#endif
// `packet_in`: the first 80 B of the hit triangle's shading packet (load_shade_packet_raw(packet_at)), fetched by the caller so that it can be
// in flight together with the caller's own fetches; `packet_at` = where the rest is.
// PRE: `pre` is the vertex's environment light sample, drawn ahead of time with this vertex's random numbers (wavefront pipeline:
// the in-place code, its LDS tables and its registers are not compiled in); otherwise it is drawn here.
template <bool PRE = false>
PT_DEV bool shade_closest_hit(const SceneRec& sc, const FrameConstants& fc, uint32_t seed, uint32_t px, uint32_t py, const Ray& ray, const HitRec& hit,
                              const RawPacket& packet_in, const ShadePacket* packet_at, PathState& ps, Followups& fu, unsigned& taps, const EnvSample* pre = nullptr,
                              const uint32_t* occ_row = nullptr) {
    const uint32_t flags = fc.flags;
    fu.add = v3(0); fu.overwrite = false; fu.counted_shadow = 0; fu.light_index = 0; fu.hint_env = fu.hint_light = 0xffffffffu;
    if (occ_row) fu.hint_env = occ_row[0];
    fu.q_env = fu.q_light = fu.q_bounce = false;
#ifdef PT_TIMING
    unsigned long long _sec[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, _t = __builtin_readcyclecounter();
#endif
    const ShadeInst inst = load_shade_inst(sc, raw_packet_inst(packet_in));
    RawPacket packet = packet_in;
    // the second UV set / vertex colours: a second fetch, only in waves that hold a hit on a mesh with either stream
    if (__any((inst.streams & (SI_TEXCOORD1 | SI_COLOR)) != 0)) {
        if (inst.streams & (SI_TEXCOORD1 | SI_COLOR)) load_shade_packet_extra(packet, packet_at);
    }
    const PacketVerts pv = unpack_shade_packet(packet);               // the three vertices and the instance id
    const RMat* mat = sc.rmats + inst.material_id;
    const MatHeader mh = material_header(sc, inst.material_id);
    HitGeom va = get_vertex_attributes(sc, inst, pv, v3(1 - hit.u - hit.v, hit.u, hit.v));
    PT_TICK(0)
    const int dbg = fc.debug_output;
    if (dbg >= PT_DEBUG_OUTPUT_HIT_KIND && dbg <= PT_DEBUG_OUTPUT_TEXCOORD_1) {                       // :806-840
        vec3 c;
        switch (dbg) {
            case PT_DEBUG_OUTPUT_HIT_KIND: c = hit.front ? v3(1, 0, 0) : v3(0, 1, 0); break;
            case PT_DEBUG_OUTPUT_VERTEX_COLOR: c = xyz(va.color); break;
            case PT_DEBUG_OUTPUT_VERTEX_ALPHA: c = v3(va.color.w); break;
            case PT_DEBUG_OUTPUT_VERTEX_NORMAL: c = (va.n + 1) / 2; break;
            case PT_DEBUG_OUTPUT_VERTEX_TANGENT: c = (va.t + 1) / 2; break;
            case PT_DEBUG_OUTPUT_VERTEX_BITANGENT: c = (va.bt + 1) / 2; break;
            case PT_DEBUG_OUTPUT_TEXCOORD_0: c = v3(va.tc[0].x, va.tc[0].y, 0); break;
            default: c = v3(va.tc[1].x, va.tc[1].y, 0); break;
        }
        fu.add = ps.beta * c;
        return true;
    }
    if (!hit.front) { va.ng = -va.ng; va.n = -va.n; va.t = -va.t; va.tw = -va.tw; }                  // :842-846 (float4 negation: w too)
    const vec3 intersection = ray.o + (ray.d * hit.t);                                               // :849
    const vec3 o_above = offset_ray(va.position, va.ng), o_below = offset_ray(va.position, -va.ng);
    const vec3 view = -normalize(ray.d);
    Surface sp = get_surface(sc, flags, mat, mh, va, view, taps);
    PT_TICK(1)
    if (dbg >= PT_DEBUG_OUTPUT_COLOR && dbg <= PT_DEBUG_OUTPUT_TRANSMISSIVE) {                       // :863-917
        vec3 c;
        switch (dbg) {
            case PT_DEBUG_OUTPUT_COLOR: c = sp.albedo; break;
            case PT_DEBUG_OUTPUT_ALPHA: c = v3(sp.alpha); break;
            case PT_DEBUG_OUTPUT_SHADING_NORMAL: c = (sp.n + 1) / 2; break;
            case PT_DEBUG_OUTPUT_SHADING_TANGENT: c = (sp.at + 1) / 2; break;
            case PT_DEBUG_OUTPUT_SHADING_BITANGENT: c = (sp.ab + 1) / 2; break;
            case PT_DEBUG_OUTPUT_METALNESS: c = v3(sp.metalness); break;
            case PT_DEBUG_OUTPUT_ROUGHNESS: c = v3(sqrtf(sp.ay)); break;
            case PT_DEBUG_OUTPUT_SPECULAR: c = v3(sp.spec_factor); break;
            case PT_DEBUG_OUTPUT_SPECULAR_COLOR: c = sp.spec_color; break;
            case PT_DEBUG_OUTPUT_CLEARCOAT: c = v3(sp.clearcoat); break;
            case PT_DEBUG_OUTPUT_CLEARCOAT_ROUGHNESS: c = v3(sp.cc_rough); break;
            case PT_DEBUG_OUTPUT_CLEARCOAT_NORMAL: c = (sp.cc_n + 1) / 2; break;
            default: c = v3(sp.transmissive); break;
        }
        fu.add = ps.beta * c;
        return true;
    }
    if (dbg == PT_DEBUG_OUTPUT_HEMISPHERE_VIEW_SIDE) {                                               // :919-922
        fu.add = ps.beta * (dot(view, sp.n) > 0 ? v3(0, 1, 0) : v3(1, 0, 0));
        return true;
    }
    const Lobes lobes = lobe_probabilities(sp, view);
    sp.sheen_lh = sp.sheen_sv = 0.0f;
    if (!PT_SHEEN_SKIP || __any(sp.sheen_color.x != 0.0f || sp.sheen_color.y != 0.0f || sp.sheen_color.z != 0.0f)) prepare_sheen(sp, view);   // (gltf_bsdf's own condition)
    vec3 c = emissive_of(sc, mat, mh, va.tc, taps, sp.emissive_texel);                                                     // :925-926
    fu.origin_above = o_above;
    PT_TICK(2)
    // environment NEE :929-942 (SampleEnvironmentMap :688-703)
    if (ps.bounce < fc.max_bounces && (flags & PT_FLAG_ENVIRONMENT_MAP) && (flags & PT_FLAG_ENVIRONMENT_MIS)) {
        EnvSample es;
        if (PRE) { es = *pre; ps.rc++; }                                    // the draw still counts
        else {
            vec4 r = next_random(px, py, seed, ps.rc);
            es = environment_light_sample(sc, fc.environment_intensity, r.x, r.y, importance_lds_top());
        }
        const float light_pdf = es.pdf;
        const vec3 ldir = es.dir, lcol = es.color;
        PT_TICK(6)                                                          // [6] the environment light sample (random numbers, descent, cube)
        vec3 contrib = v3(0);
        if (any_gt0(lcol)) {
            float bp = 0;
            vec3 f = evaluate_bsdf(flags, sc.sheen_e, sp, lobes, va.ng, view, ldir, bp);
            float mis = fdiv(light_pdf, light_pdf + bp);
            contrib = (mis * f * lcol) / light_pdf;
        }
        if (flags & PT_FLAG_INDIRECT_ENVIRONMENT_ONLY) c += contrib;                                 // TraceShadowRay returns 1 untraced (:726-728)
        else {
            fu.pend_env = ps.beta * contrib;
            // The reference traces this ray before it knows the sample is worth anything (:932).  With null-ray culling on, a
            // sample whose weighted contribution is exactly (0,0,0) is not traced: T * 0 adds nothing whatever T is (a NaN term is
            // not zero and is still traced).  Off by default so ray counts match the reference's.
            if (!(fc.cull_null_shadow && fu.pend_env.x == 0.0f && fu.pend_env.y == 0.0f && fu.pend_env.z == 0.0f)) { fu.q_env = true; fu.env_dir = ldir; }
        }
    }
    PT_TICK(3)
    // punctual-light NEE :945-956 (SamplePointLight :680-686)
    if ((flags & PT_FLAG_POINT_LIGHTS) && fc.num_of_lights > 0) {
        float u = next_random(px, py, seed, ps.rc).x;
        uint32_t li = f2u(u * (float)fc.num_of_lights);
        li = min(li, (uint32_t)(fc.num_of_lights - 1));                                              // u may be exactly 1 (quirk q17)
        fu.light_index = li;
        if (occ_row) fu.hint_light = occ_row[1u + li % 7u];
        float pdf = fdiv(1.0f, (float)fc.num_of_lights);
        vec3 ldir, lcol;
        bool cone_terms_staged;
        const pt_light light = load_light(sc, li, cone_terms_staged);
        light_ray(light, intersection, ldir, lcol, cone_terms_staged);
        vec3 contrib = v3(0);
        if (any_gt0(lcol)) {
            float bp = 0;
            vec3 f = evaluate_bsdf(flags, sc.sheen_e, sp, lobes, va.ng, view, ldir, bp);
            contrib = (lcol * f) / pdf;
        }
        if ((flags & PT_FLAG_SHADOW_RAYS) && !(flags & PT_FLAG_INDIRECT_ENVIRONMENT_ONLY)) {
            fu.pend_light = ps.beta * contrib;
            if (!(fc.cull_null_shadow && fu.pend_light.x == 0.0f && fu.pend_light.y == 0.0f && fu.pend_light.z == 0.0f)) { fu.q_light = true; fu.light_dir = ldir; }
        }
        else c += contrib;
    }
    PT_TICK(4)
    fu.add = ps.beta * c;
    // BSDF sampling + Russian roulette :958-1006
    if (ps.bounce < fc.max_bounces) {
        vec4 r = next_random(px, py, seed, ps.rc);
        bool is_tr = false, use_mis = false;
        float bp = 1;
        vec3 l = v3(0);
        vec3 f = sample_bsdf(flags, sc.sheen_e, sp, lobes, v3(r.x, r.y, r.z), view, l, bp, is_tr, use_mis);
        vec3 weight = bp != 0 ? f / bp : v3(0);
        vec3 throughput = ps.thr * weight;
        if (dbg >= PT_DEBUG_OUTPUT_BOUNCE_DIRECTION && dbg <= PT_DEBUG_BOUNCE_IS_TRANSMISSION) {     // :969-990
            vec3 dc;
            switch (dbg) {
                case PT_DEBUG_OUTPUT_BOUNCE_DIRECTION: dc = 0.5f * (l + 1); break;
                case PT_DEBUG_OUTPUT_BOUNCE_BSDF: dc = f; break;
                case PT_DEBUG_OUTPUT_BOUNCE_PDF: dc = v3(bp); break;
                case PT_DEBUG_OUTPUT_BOUNCE_WEIGHT: dc = weight; break;
                default: dc = is_tr ? v3(0, 1, 0) : v3(1, 0, 0); break;
            }
            // payload.color is OVERWRITTEN here: emissive / NEE are discarded; the reference has already traced the NEE
            // shadow rays by now, so they still count as rays.
            fu.counted_shadow = (fu.q_env ? 1u : 0u) + (fu.q_light ? 1u : 0u);
            fu.q_env = fu.q_light = false;
            fu.add = ps.beta * dc;
            fu.overwrite = true;
            return true;
        }
        if (any_gt0(throughput)) {
            float ur = next_random(px, py, seed, ps.rc).x;                                        // drawn even below min_bounces (quirk q6)
            bool cont = ps.bounce < fc.min_bounces;
            if (!cont) {                                                                             // RussianRoulette :712-722
                float p = clampf(max3(throughput), fc.min_rr, fc.max_rr);
                if (ur < p) { weight = weight / p; cont = true; }
            }
            if (cont) {
                fu.q_bounce = true;
                fu.b_o = is_tr ? o_below : o_above;
                fu.b_d = l;
                fu.b_beta = ps.beta * weight;
                fu.b_thr = throughput * weight;                                                      // weight applied twice (quirk q5)
                fu.b_pdf = bp; fu.b_mis = use_mis;
            }
        }
    }
    PT_TICK(5)
    PT_TICK_FLUSH()
    return false;
}

// RayGeneration prologue :744-758 (GenerateCameraRay :131-142)
PT_DEV Ray camera_ray(const FrameConstants& fc, uint32_t seed, uint32_t px, uint32_t py, int& rc) {
    vec4 r = next_random(px, py, seed, rc);
    float jx = r.x - 0.5f, jy = r.y - 0.5f;
    float cx = fdiv((float)px + 0.5f + jx, (float)fc.res_x) * 2 - 1;
    float cy = fdiv((float)py + 0.5f + jy, (float)fc.res_y) * 2 - 1;
    cy = -cy;
    vec4 s = mul4(fc.clip_to_world, vec4{cx, cy, 1, 1});
    vec4 e = mul4(fc.clip_to_world, vec4{cx, cy, 0, 1});
    vec3 o = xyz(s) / s.w;
    vec3 d = xyz(e) / e.w - o;
    Ray ray;
    ray.o = o; ray.tmin = 0; ray.d = normalize(d); ray.tmax = length(d);
    return ray;
}

// RayGeneration epilogue :760-785
// `accumulated` = frames already in the output when this sample is blended (SceneConstants.accumulated_frames).
PT_DEV vec3 sanitize_sample(const FrameConstants& fc, vec3 L) {
    const uint32_t flags = fc.flags;
    if (any_nan(L)) L = (flags & PT_FLAG_SHOW_NAN) ? v3(1, 0, 0) : v3(0);
    if (any_inf(L)) L = (flags & PT_FLAG_SHOW_INF) ? v3(1, 0, 0) : v3(0);
    if (flags & PT_FLAG_LUMINANCE_CLAMP) {
        float lum = luminance(L);
        if (lum > fc.luminance_clamp) L *= fdiv(fc.luminance_clamp, lum);
    }
    return L;
}
// the running mean: `h` is the pixel with `accumulated` frames in it
PT_DEV float4 blend_sample(float4 h, int accumulated, vec3 L) {
    float blend = fdiv(1.0f, (float)accumulated + 1.0f);
    return make_float4(h.x + blend * (L.x - h.x), h.y + blend * (L.y - h.y), h.z + blend * (L.z - h.z), h.w + blend * (1.0f - h.w));
}
PT_DEV void write_pixel(const FrameConstants& fc, int accumulated, float4* __restrict__ output, uint32_t px, uint32_t py, vec3 L) {
    L = sanitize_sample(fc, L);
    float4* outp = output + ((size_t)py * fc.res_x + px);
    if ((fc.flags & PT_FLAG_ACCUMULATE) && accumulated != 0) *outp = blend_sample(*outp, accumulated, L);
    else *outp = make_float4(L.x, L.y, L.z, 1.0f);
}

// tile-sharded pixel of a (rank-local) slot: slot -> (tile of this rank, lane in tile), one wave64 per 8x8 quadrant
// sample index of a slot in a sample batch (0 when spp == 1), and the seed that sample draws with
PT_DEV uint32_t fast_div(const FastDiv& f, uint32_t x) { return f.d == 1u ? x : (__umulhi(x, f.mul) >> f.shift); }      // x < 2^31 (pt_types.h FastDiv)
PT_DEV uint32_t slot_sample(const FrameConstants& fc, uint32_t slot) { return fc.spp > 1 ? fast_div(fc.div_pixel_slots, slot) : 0u; }
PT_DEV uint32_t sample_seed(const FrameConstants& fc, uint32_t sample) { return fc.seed + sample * fc.seed_step; }
PT_DEV bool slot_pixel(const FrameConstants& fc, uint32_t slot, uint32_t& px, uint32_t& py) {
    if (fc.spp > 1) slot -= fast_div(fc.div_pixel_slots, slot) * fc.pixel_slots;
    const uint32_t local_tile = slot >> 8, t = slot & 255;
    const uint32_t tile = fc.tile_rank + local_tile * fc.tile_rank_count;
    const uint32_t ty = fast_div(fc.div_tiles_x, tile), tx = tile - ty * fc.tiles_x;
    const uint32_t wave = t >> 6, lane = t & 63;
    px = tx * PT_TILE + (wave & 1) * 8 + (lane & 7);
    py = ty * PT_TILE + (wave >> 1) * 8 + (lane >> 3);
    return (px < fc.res_x) && (py < fc.res_y) && (tile < fc.tiles_x * fc.tiles_y);
}

// wave64 reduction of the per-lane tallies, one atomic per wave per counter
// occlusion: the node / triangle tallies go to Counters::nodes_shadow / tris_shadow (slots 8, 9) instead of nodes / tris (3, 4)
// the two tallies every build keeps (not only the counting instantiations): pushes a full stack dropped, pushes that went to the deep stack
PT_DEV void flush_rare(Counters* __restrict__ counters, const LaneStats& st) {
    if (st.overflow) atomicAdd(&counters->stack_overflow, (unsigned long long)st.overflow);
    if (st.deep) atomicAdd(&counters->deep_pushes, (unsigned long long)st.deep);
}
PT_DEV void flush_counters(Counters* __restrict__ counters, uint32_t lane, unsigned n_primary, unsigned n_bounce, unsigned n_shadow, unsigned n_hits,
                           const LaneStats& st, bool occlusion = false) {
    unsigned vals[8] = {n_primary, n_bounce, n_shadow, st.nodes, st.tris, n_hits, st.taps, st.overflow};
#pragma unroll
    for (int k = 0; k < 8; k++) {
        unsigned v = vals[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        const int slot = (occlusion && (k == 3 || k == 4)) ? k + 5 : k;
        if (lane == 0 && v) atomicAdd(((unsigned long long*)counters) + slot, (unsigned long long)v);
    }
}

}  // namespace pt
