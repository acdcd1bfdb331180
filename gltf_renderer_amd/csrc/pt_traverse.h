// pt_traverse.h -- software BVH traversal for gfx950 (replaces DXR TraceRay).
//
// One lane = one ray.  While-while traversal of the 64-B BVH2 nodes (4 x dwordx4 loads per node, both
// child boxes tested from registers), 48-B world-space triangle packets (3 x dwordx4), a per-lane stack
// held in LDS ([depth][lane] layout: conflict-free ds_read/ds_write_b32) with a scratch spill for the
// rare deep path.  DXR semantics kept: hit interval tmin < t < tmax, object-space facing (mirrored
// instances flip), instance cull-disable / force-non-opaque flags, any-hit for MASK instances,
// accept-first-hit occlusion rays, and the alpha-shadow transmittance product.
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
    for (;;) {
        if (cur >= 0) {
            const float4* np = nodes + (size_t)cur * 4;
            float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
            if (COUNT) st.nodes++;
            float limit = all_candidates ? r.tmax : best.t;
            // child 0: lo (n0.x n0.y n0.z) hi (n0.w n1.x n1.y); child 1: lo (n1.z n1.w n2.x) hi (n2.y n2.z n2.w)
            float a0 = n0.x * inv.x - ood.x, b0 = n0.w * inv.x - ood.x;
            float a1 = n0.y * inv.y - ood.y, b1 = n1.x * inv.y - ood.y;
            float a2 = n0.z * inv.z - ood.z, b2 = n1.y * inv.z - ood.z;
            float tn0 = fmaxf(fmaxf(fminf(a0, b0), fminf(a1, b1)), fmaxf(fminf(a2, b2), r.tmin));
            float tx0 = fminf(fminf(fmaxf(a0, b0), fmaxf(a1, b1)), fminf(fmaxf(a2, b2), limit)) * 1.0000004f;
            float c0 = n1.z * inv.x - ood.x, d0 = n2.y * inv.x - ood.x;
            float c1 = n1.w * inv.y - ood.y, d1 = n2.z * inv.y - ood.y;
            float c2 = n2.x * inv.z - ood.z, d2 = n2.w * inv.z - ood.z;
            float tn1 = fmaxf(fmaxf(fminf(c0, d0), fminf(c1, d1)), fmaxf(fminf(c2, d2), r.tmin));
            float tx1 = fminf(fminf(fmaxf(c0, d0), fmaxf(c1, d1)), fminf(fmaxf(c2, d2), limit)) * 1.0000004f;
            bool h0 = tn0 <= tx0, h1 = tn1 <= tx1;
            int ch0 = __float_as_int(n3.x), ch1 = __float_as_int(n3.y);
            if (h0 && h1) {
                bool swap = tn1 < tn0;
                int nearc = swap ? ch1 : ch0, farc = swap ? ch0 : ch1;
                if (sp < kStackLds) lds_stack[sp * kBlock] = farc;
                else if (sp < kStackLds + kStackSpill) spill[sp - kStackLds] = farc;
                else st.overflow++;
                if (sp < kStackLds + kStackSpill) sp++;
                cur = nearc;
                continue;
            }
            if (h0) { cur = ch0; continue; }
            if (h1) { cur = ch1; continue; }
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
