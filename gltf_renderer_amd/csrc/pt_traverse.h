// pt_traverse.h -- software BVH traversal for gfx950 (replaces DXR TraceRay).
//
// One lane = one ray.  Traversal of 64-B 4-wide nodes with 8-bit quantised child boxes (3 x dwordx4 + 1 x dwordx2
// loads per node; each word holds one bound of all four children, dequantised with v_cvt_f32_ubyte + v_fma, then
// the slab test of the four boxes is straight VALU on registers.  Measured: an extra dwordx4 load per node step
// costs 6 % of the frame, 110 extra VALU instructions per node step cost 1 %), children visited near-to-far (4-key sorting network on (entry distance | slot) packed in one uint), 48-B world-space
// triangle packets (3 x dwordx4), a per-lane stack held in LDS ([depth][lane] layout: conflict-free
// ds_read/ds_write_b32) with a scratch spill for the rare deep path.
//
// Two drivers over the same step functions:
//   traverse()          one ray per lane until it finishes (megakernel).
//   trace_persistent()  wave-persistent "while-while" loop for the wavefront stages: all lanes first run node
//                       steps until none has an inner node pending, then the lanes holding a leaf test their
//                       triangle; lanes whose ray finished pull a NEW ray from the shard's queue (ballot +
//                       one atomic per wave) once enough lanes are idle.  Measured before this change the
//                       traversal kernels ran with 21-28 % of lanes active (a wave lived as long as its
//                       slowest ray) while the VALU pipe was ~50 % busy: lane refill is the lever.
//
// DXR semantics kept: hit interval tmin < t < tmax, object-space facing (mirrored instances flip), instance
// cull-disable / force-non-opaque flags, any-hit for MASK instances, accept-first-hit occlusion rays, and
// the alpha-shadow transmittance product.
#pragma once
#include "pt_shading.h"

namespace pt {

#ifndef PT_OCC_SLOT_ORDER
#define PT_OCC_SLOT_ORDER 0
#endif
#ifndef PT_LEAF_SINGLE
#define PT_LEAF_SINGLE 0
#endif
#ifndef PT_STACK_LDS
#define PT_STACK_LDS 24
#endif
// Distance to a box plane.  1 (default): (plane - origin) * inv, the subtraction first: its rounding error is RELATIVE to the distance, which
// the 1.0000004 on the exit distance covers, so a box that the ray enters is never culled.  0: plane * inv - origin * inv as one fused
// multiply-add (six VALU instructions fewer per child): its error is ABSOLUTE, half an ulp of |origin * inv|, and for a short ray that starts
// far from the coordinate origin that is more than the padding -- measured at 1920x1080 on the Sponza-class scene, about one ray in ten
// million then missed a box whose triangle it hits (a wall's box has no thickness), left the scene through the wall and came back as a
// firefly: 43 pixels of a 64-sample frame beyond 1e-2 of the CPU oracle's, image metric 7.9e-4 of the 1e-3 allowed; subtracting first: 2 pixels,
// 4.0e-5, and the frame time is the same (22.7 against 22.8 ms: the traversal waits on its node loads, not on these instructions).
#ifndef PT_TIE_BREAK
#define PT_TIE_BREAK 1
#endif
#ifndef PT_BOX_GATE
#define PT_BOX_GATE 1
#endif
#ifndef PT_SLAB_SUBTRACT_FIRST
#define PT_SLAB_SUBTRACT_FIRST 1
#endif
#if PT_SLAB_SUBTRACT_FIRST
#define PT_SLAB_T(P, A) (((P) - t.o.A) * t.inv.A)
#else
#define PT_SLAB_T(P, A) ((P) * t.inv.A - t.ood.A)
#endif
constexpr int kStackLds = PT_STACK_LDS;       // entries per lane in LDS  (24 * 4 B * 256 lanes = 24 KiB per workgroup)
constexpr int kStackSpill = 40;     // further entries in scratch
constexpr int kBlock = 256;
constexpr int kTravDone = (int)0x80000000;   // `cur` value of a finished ray (leaf refs are ~tri > INT_MIN)

enum : uint32_t { RF_CULL_BACK = 1, RF_CULL_FRONT = 2, RF_FORCE_NON_OPAQUE = 4, RF_ACCEPT_FIRST = 8 };

struct Ray { vec3 o; float tmin; vec3 d; float tmax; };
struct HitRec { float t, u, v; int tri; bool front; };

struct LaneStats { unsigned nodes, tris, taps, overflow, deep; };

// Alpha of a candidate hit: AnyHit / ShadowAnyHit (PathTracer.lib.hlsl:1010-1035, 1053-1079).
PT_DEV void candidate_alpha(const SceneRec& sc, uint32_t inst, int tri, float u, float v, unsigned& taps, float& base_alpha,
                            float& alpha, float& cutoff) {
    const InstanceRec& in = sc.instances[inst];
    vec3 w = v3(1 - u - v, u, v);
    const PacketVerts pv = load_shade_packet(sc.shade + tri);
    vec4 c = fetch_vertex_color(in.p_color != nullptr, pv, w);
    vec2 tc[2] = {fetch_texcoord(in.p_texcoord[0] != nullptr, pv.uv0, w), fetch_texcoord(in.p_texcoord[1] != nullptr, pv.uv1, w)};
    base_color_alpha(sc, sc.rmats + in.gpu.material_id, tc, c, taps, base_alpha, alpha, cutoff);
}

// Per-lane traversal state.
struct Trav {
    vec3 o, d, inv, ood;
    float tmin, tmax;          // original interval
    uint32_t rf, mask;
    int mode;                  // 0 closest hit, 1 occlusion
    bool all_candidates;       // alpha-shadow rays visit every candidate of the ORIGINAL interval (quirk q12)
    int cur, sp;
    int post;                  // a leaf this lane reached but has not tested yet (kTravDone: none): trace_persistent's postponed leaf
    HitRec best;
    float transmission;        // ShadowPayload
    bool committed;
};

PT_DEV void trav_init(Trav& t, const SceneRec& sc, const Ray& r, uint32_t rf, uint32_t mask, int mode, float transmission0) {
    t.o = r.o; t.d = r.d; t.tmin = r.tmin; t.tmax = r.tmax;
    // 1 / direction by the compiler's full division (a subnormal component must not give NaN), then CLAMPED to +-1e30: the slab test below is
    // plane * inv - origin * inv, and with inv = inf (a direction component that is exactly zero: an orthographic camera looking along a
    // world axis, a mirror bounce off an axis-aligned wall) both products are infinities whose difference is NaN -- the ray then missed every
    // box whose slab it was INSIDE of, i.e. the whole scene.  A huge finite inv keeps the products finite: (plane - origin) * 1e30 is far
    // beyond any ray interval with the sign it should have, for every plane farther than |origin| * 2^-24 from the origin's coordinate.
    const float kInvMax = 1.0e30f;
    t.inv = v3(clampf(1.0f / r.d.x, -kInvMax, kInvMax), clampf(1.0f / r.d.y, -kInvMax, kInvMax), clampf(1.0f / r.d.z, -kInvMax, kInvMax));
    t.ood = v3(r.o.x * t.inv.x, r.o.y * t.inv.y, r.o.z * t.inv.z);
    t.rf = rf; t.mask = mask; t.mode = mode;
    t.all_candidates = (mode == 1) && (rf & RF_FORCE_NON_OPAQUE);
    t.best.t = r.tmax; t.best.tri = -1; t.best.u = 0; t.best.v = 0; t.best.front = true;
    t.transmission = transmission0;
    t.committed = false;
    t.sp = 0;
    t.post = kTravDone;
    t.cur = (mask == 0 || sc.num_tris == 0) ? kTravDone : sc.root;
}

// FAST: the caller has established (wave-uniformly) that no active lane can leave the LDS part of the stack in this step, so
// a push is one predicated ds_write instead of a three-way LDS / scratch / overflow branch nest.
template <bool FAST = false>
PT_DEV void trav_push(Trav& t, const SceneRec& sc, int* lds_stack, int* spill, int ref, LaneStats& st) {
    if (FAST) { lds_stack[t.sp * kBlock] = ref; t.sp++; return; }
    if (t.sp < kStackLds) lds_stack[t.sp * kBlock] = ref;
    else if (t.sp < kStackLds + kStackSpill) spill[t.sp - kStackLds] = ref;
    else if (sc.deep_entries != 0 && t.sp < kStackLds + kStackSpill + (int)sc.deep_entries) {     // (scalar test first: no deep stack in ordinary scenes)
        sc.deep_stack[(size_t)(t.sp - (kStackLds + kStackSpill)) * sc.deep_lanes + blockIdx.x * kBlock + threadIdx.x] = ref;
        st.deep++;
    }
    else st.overflow++;
    if (t.sp < kStackLds + kStackSpill + (int)sc.deep_entries) t.sp++;
}
PT_DEV void trav_pop(Trav& t, const SceneRec& sc, const int* lds_stack, const int* spill) {
    if (t.sp == 0) { t.cur = kTravDone; return; }
    t.sp--;
    // always a ds_read (a select between the LDS and the scratch address would make this a flat_load on every pop)
    int v = lds_stack[min(t.sp, kStackLds - 1) * kBlock];
    asm volatile("" : "+v"(v));                            // keep the two loads apart (the optimiser would re-merge them)
    if (t.sp >= kStackLds) v = spill[min(t.sp, kStackLds + kStackSpill - 1) - kStackLds];
    if (sc.deep_entries != 0 && t.sp >= kStackLds + kStackSpill)
        v = sc.deep_stack[(size_t)(t.sp - (kStackLds + kStackSpill)) * sc.deep_lanes + blockIdx.x * kBlock + threadIdx.x];
    t.cur = v;
}

#define PT_CSWAP(a, b) { uint32_t _lo = min(a, b), _hi = max(a, b); a = _lo; b = _hi; }

#if PT_BVH_WIDTH == 8
// One inner-node step over an 8-wide node (pt_types.h): six dwordx4 loads, eight slab tests.  Each hit child becomes a 64-bit
// (entry distance, child reference) pair; closest-hit rays enter the nearest and push the rest (a full 19-exchange sort of the
// pairs was slower still: 4146 against 4200 Mrays/s).
// On exit t.cur is the nearest hit child, or the popped entry, or kTravDone.
template <bool COUNT, bool ORDERED = true>
PT_DEV void trav_node_step(Trav& t, const SceneRec& sc, int* lds_stack, int* spill, LaneStats& st) {
#pragma clang fp contract(fast)       // box tests only decide the visiting order: fused multiply-adds here cannot change a hit (csrc/Makefile)
    const float4* np = (const float4*)sc.nodes + (size_t)t.cur * kNodeFloat4;
    const float4 hd = np[0], ca = np[1], cb = np[2], qx = np[3], qy = np[4], qz = np[5];
    if (COUNT) st.nodes++;
    const float limit = t.all_candidates ? t.tmax : t.best.t;
    const uint32_t ex = __float_as_uint(hd.w);
    const float sx = bvh_step(ex & 0xffu), sy = bvh_step((ex >> 8) & 0xffu), sz = bvh_step((ex >> 16) & 0xffu);
    const int c[8] = {__float_as_int(ca.x), __float_as_int(ca.y), __float_as_int(ca.z), __float_as_int(ca.w),
                      __float_as_int(cb.x), __float_as_int(cb.y), __float_as_int(cb.z), __float_as_int(cb.w)};
    // words: q?.x = lo of children 0-3, q?.y = lo of 4-7, q?.z = hi of 0-3, q?.w = hi of 4-7 (byte k & 3 = child k)
    const uint32_t lx[2] = {__float_as_uint(qx.x), __float_as_uint(qx.y)}, hx[2] = {__float_as_uint(qx.z), __float_as_uint(qx.w)};
    const uint32_t ly[2] = {__float_as_uint(qy.x), __float_as_uint(qy.y)}, hy[2] = {__float_as_uint(qy.z), __float_as_uint(qy.w)};
    const uint32_t lz[2] = {__float_as_uint(qz.x), __float_as_uint(qz.y)}, hz[2] = {__float_as_uint(qz.z), __float_as_uint(qz.w)};
    unsigned long long e[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int w = k >> 2, sh = 8 * (k & 3);
        const float LX = bvh_dequant((lx[w] >> sh) & 0xffu, sx, hd.x), HX = bvh_dequant((hx[w] >> sh) & 0xffu, sx, hd.x);
        const float LY = bvh_dequant((ly[w] >> sh) & 0xffu, sy, hd.y), HY = bvh_dequant((hy[w] >> sh) & 0xffu, sy, hd.y);
        const float LZ = bvh_dequant((lz[w] >> sh) & 0xffu, sz, hd.z), HZ = bvh_dequant((hz[w] >> sh) & 0xffu, sz, hd.z);
        const float a0 = PT_SLAB_T(LX, x), b0 = PT_SLAB_T(HX, x), a1 = PT_SLAB_T(LY, y), b1 = PT_SLAB_T(HY, y);
        const float a2 = PT_SLAB_T(LZ, z), b2 = PT_SLAB_T(HZ, z);
        const float tn = fmaxf(fmaxf(fminf(a0, b0), fminf(a1, b1)), fmaxf(fminf(a2, b2), t.tmin));
        const float tx = fminf(fminf(fmaxf(a0, b0), fmaxf(a1, b1)), fminf(fmaxf(a2, b2), limit)) * 1.0000004f;
        const bool hit = tn <= tx && c[k] != kEmptyChild;
        // tn >= 0, so its bit pattern orders like the float; a miss sorts behind every hit
        e[k] = hit ? (((unsigned long long)__float_as_uint(tn) << 32) | (uint32_t)c[k]) : ~0ull;
    }
    // a step pushes at most seven entries: if no active lane is within seven of the LDS part's end, every push is a plain ds_write
    const bool shallow = __ballot(t.sp + 7 > kStackLds) == 0;
    if (!ORDERED) {
        // occlusion rays accept any hit: visiting order is irrelevant, skip the sort
        int next = kTravDone;
#define PT_PUSH_UNORDERED(F)                                                                                                        \
        _Pragma("unroll") for (int k = 0; k < 8; k++)                                                                               \
            if (e[k] != ~0ull) { if (next != kTravDone) trav_push<F>(t, sc, lds_stack, spill, (int)(uint32_t)e[k], st); else next = (int)(uint32_t)e[k]; }
        if (shallow) { PT_PUSH_UNORDERED(true) } else { PT_PUSH_UNORDERED(false) }
#undef PT_PUSH_UNORDERED
        if (next != kTravDone) t.cur = next; else trav_pop(t, sc, lds_stack, spill);
        return;
    }
    // nearest child first, the other hit children pushed as they come (their order only affects how soon a later box is culled)
    unsigned long long best = e[0];
#pragma unroll
    for (int k = 1; k < 8; k++) best = e[k] < best ? e[k] : best;
    if (best != ~0ull) {
#define PT_PUSH_REST(F)                                                                                                             \
        _Pragma("unroll") for (int k = 0; k < 8; k++) if (e[k] != ~0ull && e[k] != best) trav_push<F>(t, sc, lds_stack, spill, (int)(uint32_t)e[k], st);
        if (shallow) { PT_PUSH_REST(true) } else { PT_PUSH_REST(false) }
#undef PT_PUSH_REST
        t.cur = (int)(uint32_t)best;
    } else trav_pop(t, sc, lds_stack, spill);
}
#else
// One inner-node step: t.cur >= 0 on entry; on exit t.cur is the nearest hit child, or the popped entry, or kTravDone.
template <bool COUNT, bool ORDERED = true>
PT_DEV void trav_node_step(Trav& t, const SceneRec& sc, int* lds_stack, int* spill, LaneStats& st) {
#pragma clang fp contract(fast)       // box tests only decide the visiting order: fused multiply-adds here cannot change a hit (csrc/Makefile)
    const float4* np = (const float4*)sc.nodes + (size_t)t.cur * 4;
    const float4 hd = np[0], chf = np[1], qxy = np[2];
    const float2 qz = *(const float2*)(np + 3);
    if (COUNT) st.nodes++;
#ifdef PT_PROBE_VALU          // diagnostic build only: PT_PROBE_VALU extra dependent VALU instructions per node step
    { float x = t.tmin; _Pragma("unroll") for (int i = 0; i < PT_PROBE_VALU; i++) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x)); if (x == 123.456f) st.overflow++; }
#endif
    const float limit = t.all_candidates ? t.tmax : t.best.t;
    const int c0 = __float_as_int(chf.x), c1 = __float_as_int(chf.y), c2 = __float_as_int(chf.z), c3 = __float_as_int(chf.w);
    // dequantise the four boxes (pt_types.h Bvh4Node): plane = origin + q * 2^(exp - 127), byte k of each word = child k
    const uint32_t ex = __float_as_uint(hd.w);
    const float sx = bvh_step(ex & 0xffu), sy = bvh_step((ex >> 8) & 0xffu), sz = bvh_step((ex >> 16) & 0xffu);
    const uint32_t wlx = __float_as_uint(qxy.x), whx = __float_as_uint(qxy.y), wly = __float_as_uint(qxy.z), why = __float_as_uint(qxy.w);
    const uint32_t wlz = __float_as_uint(qz.x), whz = __float_as_uint(qz.y);
#define PT_DQ4(W, S, O) make_float4(bvh_dequant((W) & 0xffu, S, O), bvh_dequant(((W) >> 8) & 0xffu, S, O), bvh_dequant(((W) >> 16) & 0xffu, S, O), bvh_dequant((W) >> 24, S, O))
    const float4 lox = PT_DQ4(wlx, sx, hd.x), hix = PT_DQ4(whx, sx, hd.x), loy = PT_DQ4(wly, sy, hd.y), hiy = PT_DQ4(why, sy, hd.y);
    const float4 loz = PT_DQ4(wlz, sz, hd.z), hiz = PT_DQ4(whz, sz, hd.z);
#undef PT_DQ4
    uint32_t key[4];
#define PT_SLAB(K, CH, LX, LY, LZ, HX, HY, HZ)                                                                            \
    {                                                                                                                     \
        float a0 = PT_SLAB_T(LX, x), b0 = PT_SLAB_T(HX, x), a1 = PT_SLAB_T(LY, y), b1 = PT_SLAB_T(HY, y);                   \
        float a2 = PT_SLAB_T(LZ, z), b2 = PT_SLAB_T(HZ, z);                                                                \
        float tn = fmaxf(fmaxf(fminf(a0, b0), fminf(a1, b1)), fmaxf(fminf(a2, b2), t.tmin));                              \
        float tx = fminf(fminf(fmaxf(a0, b0), fmaxf(a1, b1)), fminf(fmaxf(a2, b2), limit)) * 1.0000004f;                  \
        key[K] = (tn <= tx && CH != kEmptyChild) ? ((__float_as_uint(tn) & ~3u) | (uint32_t)K) : 0xffffffffu;              \
    }
    PT_SLAB(0, c0, lox.x, loy.x, loz.x, hix.x, hiy.x, hiz.x)
    PT_SLAB(1, c1, lox.y, loy.y, loz.y, hix.y, hiy.y, hiz.y)
    PT_SLAB(2, c2, lox.z, loy.z, loz.z, hix.z, hiy.z, hiz.z)
    PT_SLAB(3, c3, lox.w, loy.w, loz.w, hix.w, hiy.w, hiz.w)
#undef PT_SLAB
    // a step pushes at most three entries: if no active lane is within three of the LDS part's end, every push is a plain ds_write
    const bool shallow = __ballot(t.sp + 3 > kStackLds) == 0;
    if (!ORDERED) {
        // occlusion rays accept any hit: visiting order is irrelevant, skip the sort
        int next = kTravDone;
#if PT_OCC_SLOT_ORDER
        // the hit children in SLOT order (the builder stores them in decreasing surface area): the first is entered, the others are
        // pushed last-slot-first so that they pop in slot order too
        if (key[3] != 0xffffffffu) next = c3;
#define PT_PUSH_UNORDERED(F)                                                                                                        \
        if (key[2] != 0xffffffffu) { if (next != kTravDone) trav_push<F>(t, sc, lds_stack, spill, next, st); next = c2; }               \
        if (key[1] != 0xffffffffu) { if (next != kTravDone) trav_push<F>(t, sc, lds_stack, spill, next, st); next = c1; }               \
        if (key[0] != 0xffffffffu) { if (next != kTravDone) trav_push<F>(t, sc, lds_stack, spill, next, st); next = c0; }
#else
        if (key[0] != 0xffffffffu) next = c0;
#define PT_PUSH_UNORDERED(F)                                                                                                        \
        if (key[1] != 0xffffffffu) { if (next != kTravDone) trav_push<F>(t, sc, lds_stack, spill, c1, st); else next = c1; }           \
        if (key[2] != 0xffffffffu) { if (next != kTravDone) trav_push<F>(t, sc, lds_stack, spill, c2, st); else next = c2; }           \
        if (key[3] != 0xffffffffu) { if (next != kTravDone) trav_push<F>(t, sc, lds_stack, spill, c3, st); else next = c3; }
#endif
        if (shallow) { PT_PUSH_UNORDERED(true) } else { PT_PUSH_UNORDERED(false) }
#undef PT_PUSH_UNORDERED
        if (next != kTravDone) t.cur = next; else trav_pop(t, sc, lds_stack, spill);
        return;
    }
    // sort the 4 keys ascending (tn >= 0, so its bit pattern orders like the float); misses sink to the end
    PT_CSWAP(key[0], key[1]) PT_CSWAP(key[2], key[3]) PT_CSWAP(key[0], key[2]) PT_CSWAP(key[1], key[3]) PT_CSWAP(key[1], key[2])
    auto child_of = [&](uint32_t k) { uint32_t s = k & 3u; return s == 0 ? c0 : (s == 1 ? c1 : (s == 2 ? c2 : c3)); };
    if (key[0] != 0xffffffffu) {
#define PT_PUSH_ORDERED(F)                                                                                                          \
        if (key[3] != 0xffffffffu) trav_push<F>(t, sc, lds_stack, spill, child_of(key[3]), st);                                         \
        if (key[2] != 0xffffffffu) trav_push<F>(t, sc, lds_stack, spill, child_of(key[2]), st);                                         \
        if (key[1] != 0xffffffffu) trav_push<F>(t, sc, lds_stack, spill, child_of(key[1]), st);
        if (shallow) { PT_PUSH_ORDERED(true) } else { PT_PUSH_ORDERED(false) }
#undef PT_PUSH_ORDERED
        t.cur = child_of(key[0]);
    } else trav_pop(t, sc, lds_stack, spill);
}
#endif

// What makes the answer independent of the TREE (rare path: only for a triangle the float Moeller-Trumbore test has just accepted).
//  (1) The box gate.  The float test does not decide "inside" exactly: it accepts rays that pass a few ulp (of the ray's length, more at
//      grazing incidence) outside the triangle, hence sometimes outside the triangle's box, and whether a tree's boxes cull such a ray
//      before the triangle is asked depends on the tree -- two trees (this one, the CPU oracle's binary one) then disagree on about one ray
//      in 10^8, each a different path: fireflies.  So a candidate also has to pass the box test of ITS OWN box, in the node test's
//      arithmetic, and its distance has to be consistent with that box.  Every ancestor's box contains the triangle's (the builder checks
//      the dequantised planes against it), each operation of the box test is monotone in the plane under float rounding, so an ancestor
//      passes whenever the triangle's own box does: no tree culls a candidate that stands, and none is asked about one that does not.
//  (2) The tie rule.  Two triangles at EXACTLY the same distance (coplanar, overlapping surfaces): DXR leaves the winner to the order of the
//      walk; here the lower (instance, primitive) wins.
// The oracle states both the same way (oracle.cpp Tracer::intersect), and its exhaustive search over all triangles finds the same hits.
PT_DEV bool candidate_stands(const Trav& t, const SceneRec& sc, vec3 v0, vec3 e1, vec3 e2, float tt, float limit, uint32_t inst, uint32_t prim) {
    if (PT_BOX_GATE) {
        const vec3 v1 = v0 + e1, v2 = v0 + e2;                                          // the builder's expression for the box (accel.hip k_seg_pass)
        const vec3 lo = hmin(hmin(v0, v1), v2), hi = hmax(hmax(v0, v1), v2);
        const float a0 = (lo.x - t.o.x) * t.inv.x, b0 = (hi.x - t.o.x) * t.inv.x, a1 = (lo.y - t.o.y) * t.inv.y, b1 = (hi.y - t.o.y) * t.inv.y;
        const float a2 = (lo.z - t.o.z) * t.inv.z, b2 = (hi.z - t.o.z) * t.inv.z;
        const float tn = fmaxf(fmaxf(fminf(a0, b0), fminf(a1, b1)), fmaxf(fminf(a2, b2), t.tmin));
        const float tx = fminf(fminf(fmaxf(a0, b0), fmaxf(a1, b1)), fmaxf(a2, b2));
        if (!(tn <= tx * 1.0000004f && tn <= tt * 1.0000004f)) return false;
    }
    if (tt < limit) return true;
    if (!(PT_TIE_BREAK && t.mode == 0 && t.best.tri >= 0)) return false;                // tt == limit: the interval's end, or a tie with the hit held
    const uint4 h0 = *(const uint4*)((const float4*)sc.tris + (size_t)t.best.tri * kTriFloat4), h1 = *(const uint4*)((const float4*)sc.tris + (size_t)t.best.tri * kTriFloat4 + 1);
    return inst < h0.w || (inst == h0.w && prim < h1.w);
}

// One leaf step: t.cur = leaf reference (1..kLeafMax contiguous triangles) on entry; on exit the popped entry or kTravDone.
// `keep` != kTravDone: the leaf tested is a POSTPONED one (trace_persistent): instead of popping, the lane goes on with `keep`, the entry it
// had already moved on to -- unless the ray ended in this leaf.
template <bool COUNT>
PT_DEV void trav_leaf_step(Trav& t, const SceneRec& sc, const int* lds_stack, const int* spill, LaneStats& st, bool keep_next = false, int keep = kTravDone) {
    const uint32_t leaf = (uint32_t)~t.cur;
    const int first = (int)(leaf & kLeafFirstMask), count = (int)(leaf >> 28) + 1;
    bool stop = false;
#if PT_LEAF_SINGLE
  const int n_here = 1;                                       // one triangle per call: a lane with more of the leaf left stays at the (shortened) leaf
#else
  const int n_here = count;
#endif
  for (int k = 0; k < n_here && !stop; k++) {
    const int tri = first + k;
    const float4* tp = (const float4*)sc.tris + (size_t)tri * kTriFloat4;
    float4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
    if (COUNT) st.tris++;
    vec3 v0 = v3(q0.x, q0.y, q0.z), e1 = v3(q1.x, q1.y, q1.z), e2 = v3(q2.x, q2.y, q2.z);
    uint32_t tflags = __float_as_uint(q2.w);
    // Moeller-Trumbore, barycentrics (u, v) = weights of vertex 1 and 2
    vec3 p = cross(t.d, e2);
    float det = dot(e1, p);
    if (det != 0.0f && det == det) {
        float invd = 1.0f / det;
        vec3 tv = t.o - v0;
        float u = dot(tv, p) * invd;
        vec3 q = cross(tv, e1);
        float v = dot(t.d, q) * invd;
        float tt = dot(e2, q) * invd;
        float limit = t.all_candidates ? t.tmax : t.best.t;
        bool ok = (u >= 0.0f) && (u <= 1.0f) && (v >= 0.0f) && (u + v <= 1.0f) && (tt > t.tmin) && (tt <= limit);
        if (ok) ok = candidate_stands(t, sc, v0, e1, e2, tt, limit, __float_as_uint(q0.w), __float_as_uint(q1.w));
        if (ok && (t.mask & tflags & 0xffu)) {
            bool front = (det > 0.0f) != ((tflags & TF_MIRRORED) != 0);
            bool culled = false;
            if (!(tflags & TF_CULL_DISABLE)) culled = ((t.rf & RF_CULL_BACK) && !front) || ((t.rf & RF_CULL_FRONT) && front);
            if (!culled) {
                bool accept = true;
                if ((tflags & TF_FORCE_NON_OPAQUE) || (t.rf & RF_FORCE_NON_OPAQUE)) {
                    float base_a, a, cutoff;
                    candidate_alpha(sc, __float_as_uint(q0.w), tri, u, v, st.taps, base_a, a, cutoff);
                    if (t.mode == 0) accept = !(base_a < cutoff);                 // IgnoreHit
                    else {
                        t.transmission *= 1 - a;
                        if (t.transmission == 0.0f) stop = true;                  // AcceptHitAndEndSearch
                    }
                }
                if (accept) {
                    t.committed = true;
                    if (!t.all_candidates || tt < t.best.t) { t.best.t = tt; t.best.u = u; t.best.v = v; t.best.tri = tri; t.best.front = front; }
                    if (t.rf & RF_ACCEPT_FIRST) stop = true;
                }
            }
        }
    }
  }
    if (stop) t.cur = kTravDone;
#if PT_LEAF_SINGLE
    else if (count > 1) t.cur = ~(int)(((uint32_t)(first + 1) & kLeafFirstMask) | ((uint32_t)(count - 2) << 28));
#endif
    else if (keep_next) t.cur = keep;
    else trav_pop(t, sc, lds_stack, spill);
}

// mode 0: closest hit (hit group 0).  mode 1: occlusion / shadow (hit group 1), `transmission` is the ShadowPayload.
// Returns true if a hit was committed.
template <bool COUNT>
PT_DEV bool traverse(const SceneRec& sc, int* lds_stack, const Ray& r, uint32_t rf, uint32_t mask, int mode, HitRec& best,
                     float& transmission, LaneStats& st) {
    Trav t;
    int spill[kStackSpill];
    trav_init(t, sc, r, rf, mask, mode, transmission);
    while (t.cur != kTravDone) {
        if (t.cur >= 0) trav_node_step<COUNT>(t, sc, lds_stack, spill, st);
        else trav_leaf_step<COUNT>(t, sc, lds_stack, spill, st);
    }
    best = t.best;
    transmission = t.transmission;
    return t.committed;
}

}  // namespace pt
