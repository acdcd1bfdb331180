// pt_traverse.h -- software BVH traversal for gfx950 (replaces DXR TraceRay).
//
// One lane = one ray.  While-while traversal of 128-B 4-wide nodes (7 x dwordx4 loads per node; each float4
// holds one bound of all four children, so a slab test of the four boxes is straight VALU on registers),
// children visited near-to-far (4-key sorting network on (entry distance | slot) packed in one uint),
// 48-B world-space triangle packets (3 x dwordx4), a per-lane stack held in LDS ([depth][lane] layout:
// conflict-free ds_read/ds_write_b32) with a scratch spill for the rare deep path.  Traversal is a chain of
// dependent fetches that mostly hit L2 / Infinity Cache, i.e. latency-bound: the 4-wide node halves the
// number of dependent steps of the binary LBVH it is collapsed from.
// DXR semantics kept: hit interval tmin < t < tmax, object-space facing (mirrored instances flip), instance
// cull-disable / force-non-opaque flags, any-hit for MASK instances, accept-first-hit occlusion rays, and
// the alpha-shadow transmittance product.
#pragma once
#include "pt_shading.h"

namespace pt {

constexpr int kStackLds = 24;       // entries per lane in LDS  (24 * 4 B * 256 lanes = 24 KiB per workgroup)
constexpr int kStackSpill = 40;     // further entries in scratch
constexpr int kBlock = 256;

enum : uint32_t { RF_CULL_BACK = 1, RF_CULL_FRONT = 2, RF_FORCE_NON_OPAQUE = 4, RF_ACCEPT_FIRST = 8 };

struct Ray { vec3 o; float tmin; vec3 d; float tmax; };
struct HitRec { float t, u, v; int tri; bool front; };

struct LaneStats { unsigned nodes, tris, taps, overflow; };

// Alpha of a candidate hit: AnyHit / ShadowAnyHit (PathTracer.lib.hlsl:1010-1035, 1053-1079).
PT_DEV void candidate_alpha(const SceneRec& sc, uint32_t inst, uint32_t prim, float u, float v, unsigned& taps, float& base_alpha,
                            float& alpha, float& cutoff) {
    const pt_mesh_instance& in = sc.instances[inst].gpu;
    const pt_material& m = sc.materials[in.material_id];
    vec3 w = v3(1 - u - v, u, v);
    uint32_t vi[3];
    fetch_indices(sc, in.index_descriptor, prim, vi);
    vec4 c = fetch_vertex_color(sc, in.color_descriptor, vi, w);
    vec2 tc[2] = {fetch_texcoord(sc, in.texcoord_descriptors[0], vi, w), fetch_texcoord(sc, in.texcoord_descriptors[1], vi, w)};
    c = base_color(sc, m, tc, c, taps);
    base_alpha = c.w;
    alpha = alpha_of(m, c);
    cutoff = m.alpha_cutoff;
}

#define PT_CSWAP(a, b) { uint32_t _lo = min(a, b), _hi = max(a, b); a = _lo; b = _hi; }

// mode 0: closest hit (hit group 0).  mode 1: occlusion / shadow (hit group 1), `transmission` is the ShadowPayload.
// Returns true if a hit was committed.
template <bool COUNT>
PT_DEV bool traverse(const SceneRec& sc, int* lds_stack, const Ray& r, uint32_t rf, uint32_t mask, int mode, HitRec& best,
                     float& transmission, LaneStats& st) {
    best.t = r.tmax; best.tri = -1; best.u = 0; best.v = 0; best.front = true;
    if (mask == 0 || sc.num_tris == 0) return false;
    const vec3 inv = v3(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
    const vec3 ood = v3(r.o.x * inv.x, r.o.y * inv.y, r.o.z * inv.z);
    // alpha-shadow rays visit every candidate of the ORIGINAL interval (a DXR-conformant far-to-near order, quirk q12)
    const bool all_candidates = (mode == 1) && (rf & RF_FORCE_NON_OPAQUE);
    int spill[kStackSpill];
    int sp = 0;
    int cur = sc.root;
    bool committed = false;
    const float4* nodes = (const float4*)sc.nodes;
    const float4* tris = (const float4*)sc.tris;
    auto push = [&](int ref) {
        if (sp < kStackLds) lds_stack[sp * kBlock] = ref;
        else if (sp < kStackLds + kStackSpill) spill[sp - kStackLds] = ref;
        else st.overflow++;
        if (sp < kStackLds + kStackSpill) sp++;
    };
    for (;;) {
        if (cur >= 0) {
            const float4* np = nodes + (size_t)cur * 8;
            const float4 lox = np[0], loy = np[1], loz = np[2], hix = np[3], hiy = np[4], hiz = np[5], chf = np[6];
            if (COUNT) st.nodes++;
            const float limit = all_candidates ? r.tmax : best.t;
            const int c0 = __float_as_int(chf.x), c1 = __float_as_int(chf.y), c2 = __float_as_int(chf.z), c3 = __float_as_int(chf.w);
            uint32_t key[4];
#define PT_SLAB(K, CH, LX, LY, LZ, HX, HY, HZ)                                                                     \
    {                                                                                                           \
        float a0 = LX * inv.x - ood.x, b0 = HX * inv.x - ood.x, a1 = LY * inv.y - ood.y, b1 = HY * inv.y - ood.y; \
        float a2 = LZ * inv.z - ood.z, b2 = HZ * inv.z - ood.z;                                                  \
        float tn = fmaxf(fmaxf(fminf(a0, b0), fminf(a1, b1)), fmaxf(fminf(a2, b2), r.tmin));                    \
        float tx = fminf(fminf(fmaxf(a0, b0), fmaxf(a1, b1)), fminf(fmaxf(a2, b2), limit)) * 1.0000004f;        \
        key[K] = (tn <= tx && CH != kEmptyChild) ? ((__float_as_uint(tn) & ~3u) | (uint32_t)K) : 0xffffffffu;    \
    }
            PT_SLAB(0, c0, lox.x, loy.x, loz.x, hix.x, hiy.x, hiz.x)
            PT_SLAB(1, c1, lox.y, loy.y, loz.y, hix.y, hiy.y, hiz.y)
            PT_SLAB(2, c2, lox.z, loy.z, loz.z, hix.z, hiy.z, hiz.z)
            PT_SLAB(3, c3, lox.w, loy.w, loz.w, hix.w, hiy.w, hiz.w)
#undef PT_SLAB
            // sort the 4 keys ascending (tn >= 0, so its bit pattern orders like the float); misses sink to the end
            PT_CSWAP(key[0], key[1]) PT_CSWAP(key[2], key[3]) PT_CSWAP(key[0], key[2]) PT_CSWAP(key[1], key[3]) PT_CSWAP(key[1], key[2])
            auto child_of = [&](uint32_t k) { uint32_t s = k & 3u; return s == 0 ? c0 : (s == 1 ? c1 : (s == 2 ? c2 : c3)); };
            if (key[0] != 0xffffffffu) {
                if (key[3] != 0xffffffffu) push(child_of(key[3]));
                if (key[2] != 0xffffffffu) push(child_of(key[2]));
                if (key[1] != 0xffffffffu) push(child_of(key[1]));
                cur = child_of(key[0]);
                continue;
            }
        } else {
            const float4* tp = tris + (size_t)(~cur) * 3;
            float4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
            if (COUNT) st.tris++;
            vec3 v0 = v3(q0.x, q0.y, q0.z), e1 = v3(q1.x, q1.y, q1.z), e2 = v3(q2.x, q2.y, q2.z);
            uint32_t tflags = __float_as_uint(q2.w);
            // Moeller-Trumbore, barycentrics (u, v) = weights of vertex 1 and 2
            vec3 p = cross(r.d, e2);
            float det = dot(e1, p);
            if (det != 0.0f && det == det) {
                float invd = 1.0f / det;
                vec3 tv = r.o - v0;
                float u = dot(tv, p) * invd;
                vec3 q = cross(tv, e1);
                float v = dot(r.d, q) * invd;
                float tt = dot(e2, q) * invd;
                float limit = all_candidates ? r.tmax : best.t;
                bool ok = (u >= 0.0f) && (u <= 1.0f) && (v >= 0.0f) && (u + v <= 1.0f) && (tt > r.tmin) && (tt < limit);
                if (ok && (mask & tflags & 0xffu)) {
                    bool front = (det > 0.0f) != ((tflags & TF_MIRRORED) != 0);
                    bool culled = false;
                    if (!(tflags & TF_CULL_DISABLE)) culled = ((rf & RF_CULL_BACK) && !front) || ((rf & RF_CULL_FRONT) && front);
                    if (!culled) {
                        bool accept = true, stop = false;
                        if ((tflags & TF_FORCE_NON_OPAQUE) || (rf & RF_FORCE_NON_OPAQUE)) {
                            float base_a, a, cutoff;
                            candidate_alpha(sc, __float_as_uint(q0.w), __float_as_uint(q1.w), u, v, st.taps, base_a, a, cutoff);
                            if (mode == 0) accept = !(base_a < cutoff);                 // IgnoreHit
                            else {
                                transmission *= 1 - a;
                                if (transmission == 0.0f) stop = true;                  // AcceptHitAndEndSearch
                            }
                        }
                        if (accept) {
                            committed = true;
                            if (!all_candidates || tt < best.t) { best.t = tt; best.u = u; best.v = v; best.tri = ~cur; best.front = front; }
                            if (rf & RF_ACCEPT_FIRST) stop = true;
                            if (stop) break;
                        }
                    }
                }
            }
        }
        if (sp == 0) break;
        sp--;
        cur = sp < kStackLds ? lds_stack[sp * kBlock] : spill[sp - kStackLds];
    }
    return committed;
}

}  // namespace pt
