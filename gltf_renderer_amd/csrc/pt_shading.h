// pt_shading.h -- device shading library of the gfx950 path tracer.
//
// Computes what the reference's shading headers compute (cited per function, file:line relative to the
// reference root), arranged for an iterative register-resident path state instead of DXR recursion:
// per hit the lobe probabilities are formed once and shared by the environment-NEE, light-NEE and
// BSDF-sampling evaluations; texture filtering and cube-map filtering are software (CDNA4 has no
// texture units reachable from HIP).
#pragma once
#include "pt_math.h"
#include "pt_types.h"

namespace pt {

constexpr float kMinRoughness = 0.001f;    // Bsdf.hlsli:26

// ---------------------------------------------------------------- RNG (Random.hlsli:17-30, PathTracer.lib.hlsl:144-148)
PT_DEV vec4 next_random(uint32_t px, uint32_t py, uint32_t seed, int& count) {
    uint32_t x = px * 1664525u + 1013904223u, y = py * 1664525u + 1013904223u;
    uint32_t z = seed * 1664525u + 1013904223u, w = (uint32_t)count * 1664525u + 1013904223u;
    count++;
    x += y * w; y += z * x; z += x * y; w += y * z;
    x ^= x >> 16; y ^= y >> 16; z ^= z >> 16; w ^= w >> 16;
    x += y * w; y += z * x; z += x * y; w += y * z;
    const float s = 2.3283064365386963e-10f;      // 2^-32: the fp32 value of the literal 4294967295.0 is 2^32 (quirk q29)
    return {(float)x * s, (float)y * s, (float)z * s, (float)w * s};
}

// ---------------------------------------------------------------- Vertex.hlsli / Common.hlsli
PT_DEV vec3 decode_octahedral(float ex, float ey) {                // Common.hlsli:90-103
    float z = 1.f - fabsf(ex) - fabsf(ey);
    float x = ex, y = ey;
    if (!(z >= 0.f)) {
        x = (ex >= 0 ? 1.f : -1.f) * (1.f - fabsf(ey));
        y = (ey >= 0 ? 1.f : -1.f) * (1.f - fabsf(ex));
    }
    return normalize(v3(x, y, z));
}
PT_DEV vec2 encode_octahedral(vec3 n) {                            // Common.hlsli:76-88
    float s = fabsf(n.x) + fabsf(n.y) + fabsf(n.z);
    const float rs_ = frcp_refined(s);
    float ox = fdiv_with(n.x, s, rs_), oy = fdiv_with(n.y, s, rs_), oz = fdiv_with(n.z, s, rs_);
    if (oz >= 0.f) return {ox, oy};
    return {(ox >= 0 ? 1.f : -1.f) * (1.f - fabsf(oy)), (oy >= 0 ? 1.f : -1.f) * (1.f - fabsf(ox))};
}
PT_DEV void basis_accurate(vec3 n, vec3& b1, vec3& b2) {           // Common.hlsli:46-53
    float sg = n.z >= 0.0f ? 1.0f : -1.0f;
    float a = fdiv(-1.0f, sg + n.z);
    float b = n.x * n.y * a;
    b1 = v3(1.0f + sg * n.x * n.x * a, sg * b, -sg * n.x);
    b2 = v3(b, sg + n.y * n.y * a, -n.y);
}
PT_DEV void basis_simple(vec3 n, vec3& t, vec3& b) {               // Common.hlsli:33-42
    if (fabsf(n.x) > fabsf(n.z)) b = v3(-n.y, n.x, 0);
    else b = v3(0, -n.z, n.y);
    b = normalize(b);
    t = cross(b, n);
}
// R10G10B10A2_UNORM fetch + DecodeTangentSpace (Vertex.hlsli:5-19, 46-50).  Tangent comes out negated (quirk q25).
// The tangent's angle has only 1024 values: (sin, cos) of kTau * k / 1023.  The path-tracing kernels read them from a table that
// k_tangent_lut (pt_kernel.hip) fills with this very function at context creation -- same bits, two full-precision
// transcendentals per vertex fewer in the shade stage.
PT_DEV float2 tangent_sincos_compute(uint32_t k) {
    const float angle = kTau * unorm_div<1023>((float)k);
    float sn_, cs_;
    pt_sincos(angle, sn_, cs_);
    return make_float2(sn_, cs_);
}
PT_DEV void decode_tangent_space(uint32_t p, float2 sincos, vec3& normal, vec3& tangent, float& winding) {
    float ex = unorm_div<1023>((float)(p & 0x3ff)), ey = unorm_div<1023>((float)((p >> 10) & 0x3ff));
    normal = decode_octahedral(ex * 2 - 1, ey * 2 - 1);
    vec3 ct, cb;
    basis_accurate(normal, ct, cb);
    tangent = sincos.y * ct + sincos.x * cb;
    winding = (p >> 30) != 0 ? 1.f : -1.f;               // encoded.w > 0, encoded.w = (p >> 30) / 3
}
PT_DEV void decode_tangent_space(uint32_t p, vec3& normal, vec3& tangent, float& winding) {
    decode_tangent_space(p, tangent_sincos_compute((p >> 20) & 0x3ffu), normal, tangent, winding);
}
PT_DEV uint32_t encode_tangent_space(vec3 normal, vec3 tangent, float winding) {    // Vertex.hlsli:21-44
    vec2 e = encode_octahedral(normal);
    uint32_t qx = f2u(clampf(0.5f * e.x + 0.5f, 0, 1) * 1023 + 0.5f), qy = f2u(clampf(0.5f * e.y + 0.5f, 0, 1) * 1023 + 0.5f);
    vec3 nq = decode_octahedral(2.0f * unorm_div<1023>((float)qx) - 1.0f, 2.0f * unorm_div<1023>((float)qy) - 1.0f);
    vec3 ct, cb;
    basis_accurate(nq, ct, cb);
    float angle = pt_atan2(dot(tangent, cb), dot(tangent, ct));
    uint32_t qt = f2u((fdiv(angle, kTau) + 0.5f) * 1023 + 0.5f);
    uint32_t qw = winding == 1 ? 3u : 0u;
    return qx | (qy << 10) | (qt << 20) | (qw << 30);
}

// ---------------------------------------------------------------- software texture unit
PT_DEV int wrap_addr(int i, int n, int mode) {
    const bool pow2 = (n & (n - 1)) == 0;                  // two's-complement AND is the non-negative modulo (no ~30-instruction idiv)
    if (mode == PT_ADDRESS_WRAP) { if (pow2) return i & (n - 1); int m = i % n; return m < 0 ? m + n : m; }
    if (mode == PT_ADDRESS_MIRROR) {
        int p = 2 * n, m;
        if (pow2) m = i & (p - 1); else { m = i % p; if (m < 0) m += p; }
        return m < n ? m : p - 1 - m;
    }
    return i < 0 ? 0 : (i >= n ? n - 1 : i);
}
// The sRGB decode table is gathered with lane-random indices 12+ times per hit.  Kernels built with PT_LUT_LDS stage it into
// LDS once per workgroup (stage_luts), where a random 4-byte gather costs a few cycles instead of one L1 line per distinct index.
#ifdef PT_LUT_LDS
static __shared__ float pt_lds_lut[512];               // [0,256) sRGB decode, [256,512) sheen directional albedo 16x16
PT_DEV void stage_luts(const SceneRec& sc) {           // 256-thread workgroups
    pt_lds_lut[threadIdx.x & 255u] = sc.srgb_lut[threadIdx.x & 255u];
    pt_lds_lut[256u + (threadIdx.x & 255u)] = sc.sheen_e[threadIdx.x & 255u];
    __syncthreads();
}
PT_DEV float srgb_decode(const float*, uint32_t i) { return pt_lds_lut[i]; }
PT_DEV float sheen_entry(const float*, int i) { return pt_lds_lut[256 + i]; }
static __shared__ float2 pt_lds_tangent[1024];         // (sin, cos) of the packed tangent angle; staged by the shade stage only
PT_DEV void stage_tangent_lut(const SceneRec& sc) {    // 256-thread workgroups
    for (uint32_t i = threadIdx.x; i < 1024u; i += 256u) pt_lds_tangent[i] = sc.tangent_lut[i];
    __syncthreads();
}
PT_DEV float2 tangent_sincos(const SceneRec&, uint32_t k) { return pt_lds_tangent[k]; }
#else
PT_DEV void stage_luts(const SceneRec&) {}
PT_DEV float srgb_decode(const float* lut, uint32_t i) { return lut[i]; }
PT_DEV float sheen_entry(const float* lut, int i) { return lut[i]; }
PT_DEV void stage_tangent_lut(const SceneRec&) {}
PT_DEV float2 tangent_sincos(const SceneRec& sc, uint32_t k) { return gload_f2(sc.tangent_lut + k); }
#endif
PT_DEV vec4 unpack_texel(uint32_t t, uint32_t srgb, const float* lut) {
    uint32_t r = t & 0xff, g = (t >> 8) & 0xff, b = (t >> 16) & 0xff, a = t >> 24;
    if (srgb) return {srgb_decode(lut, r), srgb_decode(lut, g), srgb_decode(lut, b), unorm_div<255>((float)a)};
    return {unorm_div<255>((float)r), unorm_div<255>((float)g), unorm_div<255>((float)b), unorm_div<255>((float)a)};
}
PT_DEV float finite_coord(float x) {
    if (!(x == x) || isinf(x)) return 0.f;
    return clampf(x, -1.0e9f, 1.0e9f);
}
PT_DEV RTex load_rtex(const RTex* p) {                      // 3 x dwordx4, issued together
    const float4* q = (const float4*)p;
    const float4 a = q[0], b = q[1], c = q[2];
    RTex t;
    t.texels = (const uint32_t*)(((uint64_t)__float_as_uint(a.y) << 32) | (uint64_t)__float_as_uint(a.x));
    t.width = __float_as_int(a.z); t.height = __float_as_int(a.w);
    t.flags = __float_as_uint(b.x); t.m00 = b.y; t.m01 = b.z; t.ox = b.w;
    t.m10 = c.x; t.m11 = c.y; t.oy = c.z; t._pad = 0;
    return t;
}
// The four texel addresses + weights of one Texture2D.SampleLevel(sampler, uv, 0): D3D texel-centre rule.
// TransformUv (Material.hlsli:68-88) is folded in: rows (c*sx, s*sy, ox), (-s*sx, c*sy, oy) were formed on the host in fp32.
#ifndef PT_TEX_PAIRS
#define PT_TEX_PAIRS 1      // the two texels of a row in ONE 8-byte load (0: four dword gathers per fetch)
#endif
#ifndef PT_TEX_TRIO
#define PT_TEX_TRIO 1       // materials flagged RM_TRIO fetch albedo + normal + metal-rough from the interleaved copy (0: never; needs PT_TEX_PAIRS)
#endif
#if PT_TEX_PAIRS
// Both texels of a row come from one 8-byte load at the pair's base column ia = clamp(i0, 0, width - 2): element i - ia of it.  Only
// a tap whose second column wrapped or mirrored off the pair (i0 = width - 1 with WRAP) needs its own loads, and the whole wave
// skips those unless some lane is there.  (Textures are allocated 4 bytes long so that a width-1 texture's pair stays inside.)
struct TexTaps { const uint32_t *p0, *p1; int i0, i1, ia, width; float w00, w10, w01, w11; uint32_t srgb; bool edge; };
struct TexQuad { uint32_t t00, t10, t01, t11; };
PT_DEV TexTaps texture_taps(const RTex& t, const vec2 tc[2]) {
    // No fused multiply-add in here: the compiler contracts each inlined copy of this function on its own, and two textures with the
    // same transform and size must get the same texels and weights to the bit (a material's interleaved footprint is fetched with the
    // albedo texture's taps).  Plain products and sums are also what the CPU oracle computes.
#pragma clang fp contract(off)
    const vec2 uv = (t.flags & RT_TEXCOORD1) ? tc[1] : tc[0];
    const float tu = t.m00 * uv.x + t.m01 * uv.y + t.ox;
    const float tv = t.m10 * uv.x + t.m11 * uv.y + t.oy;
    const int au = (int)((t.flags >> 1) & 3u), av = (int)((t.flags >> 3) & 3u);
    float x = finite_coord(tu * (float)t.width), y = finite_coord(tv * (float)t.height);
    TexTaps k;
    k.srgb = t.flags & RT_SRGB;
    k.width = t.width;
    int j0, j1;
    if (t.flags & RT_POINT) {
        k.i0 = k.i1 = wrap_addr((int)floorf(x), t.width, au); j0 = j1 = wrap_addr((int)floorf(y), t.height, av);
        k.w00 = 1; k.w10 = k.w01 = k.w11 = 0;
    } else {
        x -= 0.5f; y -= 0.5f;
        const float fx0 = floorf(x), fy0 = floorf(y);
        const float fx = x - fx0, fy = y - fy0;
        k.i0 = wrap_addr((int)fx0, t.width, au); k.i1 = wrap_addr((int)fx0 + 1, t.width, au);
        j0 = wrap_addr((int)fy0, t.height, av); j1 = wrap_addr((int)fy0 + 1, t.height, av);
        k.w00 = (1 - fx) * (1 - fy); k.w10 = fx * (1 - fy); k.w01 = (1 - fx) * fy; k.w11 = fx * fy;
    }
    k.ia = max(min(k.i0, t.width - 2), 0);
    k.edge = k.i1 != k.ia && k.i1 != k.ia + 1;
    k.p0 = t.texels + (size_t)j0 * t.width;
    k.p1 = t.texels + (size_t)j1 * t.width;
#ifdef PT_PROBE_NO_TEXELS     // PROBE ONLY (tools/pmc_shade_attribution.sh): every footprint is the texture's first texels -- wrong colours, what the texel gathers cost
    k.i0 = 0; k.i1 = min(1, t.width - 1); k.ia = 0; k.edge = false; k.p0 = t.texels; k.p1 = t.texels;
#endif
    return k;
}
PT_DEV uint2 tap_row(const TexTaps& k, int row) { return gload_u2((row ? k.p1 : k.p0) + k.ia); }
PT_DEV TexQuad tap_quad(const TexTaps& k, const uint2 r0, const uint2 r1) {
    TexQuad q;
    q.t00 = k.i0 == k.ia ? r0.x : r0.y; q.t01 = k.i0 == k.ia ? r1.x : r1.y;
    q.t10 = k.i1 == k.ia ? r0.x : r0.y; q.t11 = k.i1 == k.ia ? r1.x : r1.y;
    if (__any(k.edge)) {
        if (k.edge) { q.t10 = gload(k.p0 + k.i1); q.t11 = gload(k.p1 + k.i1); }
    }
    return q;
}
PT_DEV vec4 resolve_taps(const TexTaps& k, uint32_t t00, uint32_t t10, uint32_t t01, uint32_t t11, const float* srgb_lut) {
    if (k.w10 == 0 && k.w01 == 0 && k.w11 == 0 && k.w00 == 1) return unpack_texel(t00, k.srgb, srgb_lut);      // point filter
    return unpack_texel(t00, k.srgb, srgb_lut) * k.w00 + unpack_texel(t10, k.srgb, srgb_lut) * k.w10 +
           unpack_texel(t01, k.srgb, srgb_lut) * k.w01 + unpack_texel(t11, k.srgb, srgb_lut) * k.w11;
}
PT_DEV vec4 resolve_taps(const TexTaps& k, const TexQuad& q, const float* srgb_lut) { return resolve_taps(k, q.t00, q.t10, q.t01, q.t11, srgb_lut); }
#else
struct TexTaps { const uint32_t *p00, *p10, *p01, *p11; float w00, w10, w01, w11; uint32_t srgb; };
PT_DEV TexTaps texture_taps(const RTex& t, const vec2 tc[2]) {
    // No fused multiply-add in here: the compiler contracts each inlined copy of this function on its own, and two textures with the
    // same transform and size must get the same texels and weights to the bit (a material's interleaved footprint is fetched with the
    // albedo texture's taps).  Plain products and sums are also what the CPU oracle computes.
#pragma clang fp contract(off)
    const vec2 uv = (t.flags & RT_TEXCOORD1) ? tc[1] : tc[0];
    const float tu = t.m00 * uv.x + t.m01 * uv.y + t.ox;
    const float tv = t.m10 * uv.x + t.m11 * uv.y + t.oy;
    const int au = (int)((t.flags >> 1) & 3u), av = (int)((t.flags >> 3) & 3u);
    float x = finite_coord(tu * (float)t.width), y = finite_coord(tv * (float)t.height);
    TexTaps k;
    k.srgb = t.flags & RT_SRGB;
    if (t.flags & RT_POINT) {
        int i = wrap_addr((int)floorf(x), t.width, au), j = wrap_addr((int)floorf(y), t.height, av);
        k.p00 = k.p10 = k.p01 = k.p11 = t.texels + (size_t)j * t.width + i;
        k.w00 = 1; k.w10 = k.w01 = k.w11 = 0;
        return k;
    }
    x -= 0.5f; y -= 0.5f;
    float fx0 = floorf(x), fy0 = floorf(y);
    float fx = x - fx0, fy = y - fy0;
    int i0 = wrap_addr((int)fx0, t.width, au), i1 = wrap_addr((int)fx0 + 1, t.width, au);
    int j0 = wrap_addr((int)fy0, t.height, av), j1 = wrap_addr((int)fy0 + 1, t.height, av);
    const uint32_t* r0 = t.texels + (size_t)j0 * t.width;
    const uint32_t* r1 = t.texels + (size_t)j1 * t.width;
    k.p00 = r0 + i0; k.p10 = r0 + i1; k.p01 = r1 + i0; k.p11 = r1 + i1;
    k.w00 = (1 - fx) * (1 - fy); k.w10 = fx * (1 - fy); k.w01 = (1 - fx) * fy; k.w11 = fx * fy;
    return k;
}
PT_DEV vec4 resolve_taps(const TexTaps& k, uint32_t t00, uint32_t t10, uint32_t t01, uint32_t t11, const float* srgb_lut) {
    if (k.w10 == 0 && k.w01 == 0 && k.w11 == 0 && k.w00 == 1) return unpack_texel(t00, k.srgb, srgb_lut);      // point filter
    return unpack_texel(t00, k.srgb, srgb_lut) * k.w00 + unpack_texel(t10, k.srgb, srgb_lut) * k.w10 +
           unpack_texel(t01, k.srgb, srgb_lut) * k.w01 + unpack_texel(t11, k.srgb, srgb_lut) * k.w11;
}
#endif
// SampleTexture (Material.hlsli:90-96) of one material slot.
PT_DEV vec4 sample_slot(const SceneRec& sc, const RMat* m, int slot, const vec2 tc[2], unsigned& taps) {
    const RTex t = load_rtex(&m->tex[slot]);
    const TexTaps k = texture_taps(t, tc);
    taps++;
#if PT_TEX_PAIRS
    const uint2 r0 = tap_row(k, 0), r1 = tap_row(k, 1);
    return resolve_taps(k, tap_quad(k, r0, r1), sc.srgb_lut);
#else
    return resolve_taps(k, gload(k.p00), gload(k.p10), gload(k.p01), gload(k.p11), sc.srgb_lut);
#endif
}

// ---------------------------------------------------------------- vertex fetch (PathTracer.lib.hlsl:176-302)
// The three vertices of a hit triangle from its 128-B shading packet (pt_types.h ShadePacket: one cache line, 8 x dwordx4).
struct PacketVerts { vec3 p[3]; uint32_t ts[3]; float2 uv0[3], uv1[3]; uint2 col[3]; uint32_t inst; };
struct RawPacket { float4 r[8]; };                          // the loads, so that a caller can issue them ahead of their use
// the 80 B every hit needs (pt_types.h ShadePacket): five loads; the rest of the packet reads as zeros until load_shade_packet_extra
PT_DEV RawPacket load_shade_packet_raw(const ShadePacket* pk) {
    const float4* q = (const float4*)pk;
    RawPacket p;
#pragma unroll
    for (int k = 0; k < 5; k++) p.r[k] = q[k];
#pragma unroll
    for (int k = 5; k < 8; k++) p.r[k] = make_float4(0, 0, 0, 0);
    return p;
}
PT_DEV uint32_t raw_packet_inst(const RawPacket& p) { return __float_as_uint(p.r[4].z); }
// the second UV set and the vertex colours, for the meshes that have them
PT_DEV void load_shade_packet_extra(RawPacket& p, const ShadePacket* pk) {
    const float4* q = (const float4*)pk;
#pragma unroll
    for (int k = 5; k < 8; k++) p.r[k] = q[k];
}
PT_DEV PacketVerts unpack_shade_packet(const RawPacket& raw) {
    const float4* r = raw.r;
    PacketVerts o;
#pragma unroll
    for (int k = 0; k < 3; k++) { o.p[k] = v3(r[k].x, r[k].y, r[k].z); o.ts[k] = __float_as_uint(r[k].w); }
    o.uv0[0] = make_float2(r[3].x, r[3].y); o.uv0[1] = make_float2(r[3].z, r[3].w); o.uv0[2] = make_float2(r[4].x, r[4].y);
    o.inst = __float_as_uint(r[4].z);
    o.uv1[0] = make_float2(r[5].x, r[5].y); o.uv1[1] = make_float2(r[5].z, r[5].w); o.uv1[2] = make_float2(r[6].x, r[6].y);
    o.col[0] = make_uint2(__float_as_uint(r[6].z), __float_as_uint(r[6].w));
    o.col[1] = make_uint2(__float_as_uint(r[7].x), __float_as_uint(r[7].y));
    o.col[2] = make_uint2(__float_as_uint(r[7].z), __float_as_uint(r[7].w));
    return o;
}
PT_DEV PacketVerts load_shade_packet(const ShadePacket* pk) { RawPacket p = load_shade_packet_raw(pk); load_shade_packet_extra(p, pk); return unpack_shade_packet(p); }   // the whole packet (any-hit alpha tests)
PT_DEV vec4 fetch_vertex_color(bool present, const PacketVerts& pv, vec3 w) {                         // :229-242
    if (!present) return {1, 1, 1, 1};
    const uint2 q0 = pv.col[0], q1 = pv.col[1], q2 = pv.col[2];
    auto un = [](uint2 q) { return vec4{unorm_div<65535>((float)(q.x & 0xffff)), unorm_div<65535>((float)(q.x >> 16)), unorm_div<65535>((float)(q.y & 0xffff)), unorm_div<65535>((float)(q.y >> 16))}; };
    return un(q0) * w.x + un(q1) * w.y + un(q2) * w.z;
}
PT_DEV vec2 fetch_texcoord(bool present, const float2 t[3], vec3 w) {                                // :244-257
    if (!present) return {0, 0};
    const float2 a = t[0], b = t[1], c = t[2];
    return {w.x * a.x + w.y * b.x + w.z * c.x, w.x * a.y + w.y * b.y + w.z * c.y};
}
struct HitGeom {                       // VertexAttributes, PathTracer.lib.hlsl:270-278
    vec3 position, ng, n, t, bt;
    float tw;
    vec4 color;
    vec2 tc[2];
};
// What a hit needs of its instance row: 96 B instead of the 240-B InstanceRec, so that the rows of up to kInstCacheMax instances
// fit in LDS in the shade stage and the instance is no longer a second dependent global fetch behind the shading packet.
struct ShadeInst {
    float T[16], N[16];                // transform / normal_transform in their column-major slots (only the 12 / 9 used entries are set)
    uint32_t material_id, streams;     // streams: SI_* presence bits
};
enum : uint32_t { SI_TANGENT_SPACE = 1, SI_TEXCOORD0 = 2, SI_TEXCOORD1 = 4, SI_COLOR = 8 };
PT_DEV ShadeInst shade_inst_unpack(const float4 q[6]) {
    ShadeInst si;
    si.T[0] = q[0].x; si.T[1] = q[0].y; si.T[2] = q[0].z; si.T[12] = q[0].w;
    si.T[4] = q[1].x; si.T[5] = q[1].y; si.T[6] = q[1].z; si.T[13] = q[1].w;
    si.T[8] = q[2].x; si.T[9] = q[2].y; si.T[10] = q[2].z; si.T[14] = q[2].w;
    si.N[0] = q[3].x; si.N[1] = q[3].y; si.N[2] = q[3].z; si.material_id = __float_as_uint(q[3].w);
    si.N[4] = q[4].x; si.N[5] = q[4].y; si.N[6] = q[4].z; si.streams = __float_as_uint(q[4].w);
    si.N[8] = q[5].x; si.N[9] = q[5].y; si.N[10] = q[5].z;
    return si;
}
PT_DEV void shade_inst_pack(const InstanceRec& in, float4 q[6]) {
    const float* T = in.gpu.transform; const float* N = in.gpu.normal_transform;
    const uint32_t streams = (in.p_tangent_space ? SI_TANGENT_SPACE : 0u) | (in.p_texcoord[0] ? SI_TEXCOORD0 : 0u) | (in.p_texcoord[1] ? SI_TEXCOORD1 : 0u) |
                             (in.p_color ? SI_COLOR : 0u);
    q[0] = make_float4(T[0], T[1], T[2], T[12]); q[1] = make_float4(T[4], T[5], T[6], T[13]); q[2] = make_float4(T[8], T[9], T[10], T[14]);
    q[3] = make_float4(N[0], N[1], N[2], __uint_as_float((uint32_t)in.gpu.material_id)); q[4] = make_float4(N[4], N[5], N[6], __uint_as_float(streams));
    q[5] = make_float4(N[8], N[9], N[10], 0.0f);
}
#ifdef PT_LUT_LDS
constexpr uint32_t kInstCacheMax = 128;
static __shared__ float4 pt_lds_inst[kInstCacheMax * 6];
PT_DEV void stage_instances(const SceneRec& sc) {          // 256-thread workgroups
    const uint32_t n = sc.n_instances < kInstCacheMax ? sc.n_instances : kInstCacheMax;
    if (threadIdx.x < n) {
        float4 q[6];
        shade_inst_pack(sc.instances[threadIdx.x], q);
#pragma unroll
        for (int k = 0; k < 6; k++) pt_lds_inst[threadIdx.x * 6u + k] = q[k];
    }
    __syncthreads();
}
PT_DEV ShadeInst load_shade_inst(const SceneRec& sc, uint32_t id) {
    float4 q[6];
    if (sc.small_tables || id < kInstCacheMax) {
#pragma unroll
        for (int k = 0; k < 6; k++) q[k] = pt_lds_inst[id * 6u + k];
    } else shade_inst_pack(sc.instances[id], q);
    return shade_inst_unpack(q);
}
#else
PT_DEV void stage_instances(const SceneRec&) {}
PT_DEV ShadeInst load_shade_inst(const SceneRec& sc, uint32_t id) { float4 q[6]; shade_inst_pack(sc.instances[id], q); return shade_inst_unpack(q); }
#endif
PT_DEV HitGeom get_vertex_attributes(const SceneRec& sc, const ShadeInst& in, const PacketVerts& pv, vec3 w) {   // :280-302
    HitGeom a;
    const vec3 p0 = pv.p[0], p1 = pv.p[1], p2 = pv.p[2];
    const bool has_ts = (in.streams & SI_TANGENT_SPACE) != 0;
    const uint32_t ts0 = pv.ts[0], ts1 = pv.ts[1], ts2 = pv.ts[2];
    a.color = fetch_vertex_color((in.streams & SI_COLOR) != 0, pv, w);
    a.tc[0] = fetch_texcoord((in.streams & SI_TEXCOORD0) != 0, pv.uv0, w);
    a.tc[1] = fetch_texcoord((in.streams & SI_TEXCOORD1) != 0, pv.uv1, w);
    vec3 pos = w.x * p0 + w.y * p1 + w.z * p2;
    vec3 ng = cross(p1 - p0, p2 - p0);                     // :196-199 un-normalised
    vec3 n, t;
    float tw;
    if (has_ts) {                                          // :201-222
        vec3 n0, n1, n2, t0, t1, t2;
        float w0, w1, w2;
        decode_tangent_space(ts0, tangent_sincos(sc, (ts0 >> 20) & 0x3ffu), n0, t0, w0);
        decode_tangent_space(ts1, tangent_sincos(sc, (ts1 >> 20) & 0x3ffu), n1, t1, w1);
        decode_tangent_space(ts2, tangent_sincos(sc, (ts2 >> 20) & 0x3ffu), n2, t2, w2);
        n = w.x * n0 + w.y * n1 + w.z * n2;
        t = w.x * t0 + w.y * t1 + w.z * t2;
        tw = w0;                                           // winding from vertex 0 only (quirk q16)
    } else {
        n = ng;
        vec3 helper = v3(1, 0, 0);                         // GenerateTangent :166-174
        if (fabsf(ng.x) > fabsf(ng.y)) helper = v3(0, 1, 0);
        t = normalize(cross(helper, ng));
        tw = 1;
    }
    a.position = mul_point(in.T, pos);
    a.ng = normalize(mul_dir(in.N, ng));
    a.n = normalize(mul_dir(in.N, n));
    a.t = normalize(mul_dir(in.T, t));
    a.tw = tw;
    a.bt = tw * normalize(cross(a.n, a.t));                // :224-227
    return a;
}

// ---------------------------------------------------------------- material evaluation (Material.hlsli, PathTracer.lib.hlsl:318-381)
struct Surface {                       // live subset of SurfaceProperties (Bsdf.hlsli:4-24)
    vec3 albedo; float alpha, metalness, ax, ay;          // roughness_squared = (ax, ay)
    vec3 n, at, ab; float ior;
    vec3 spec_color; float spec_factor, clearcoat, cc_rough;
    vec3 cc_n, sheen_color; float sheen_a, transmissive;
    vec3 emissive_texel;                                  // RM_TRIO_EMISSIVE materials: the filtered emissive texel, fetched with the PBR footprint
    float sheen_lh, sheen_sv;                             // SheenL(alpha, 1/2) and SheenShadowing(alpha, n.v): the same in every evaluation at this vertex (prepare_sheen)
};
struct MatHeader {                     // the 128-B head of RMat in registers (8 x dwordx4 issued together)
    uint32_t flags; int32_t alpha_mode; float metalness_factor, roughness_factor;
    vec4 base_color_factor;
    vec3 emissive_factor; float alpha_cutoff;
    float ior, normal_scale, specular_factor, clearcoat_normal_scale;
    vec3 specular_color_factor; float clearcoat_factor;
    float clearcoat_roughness_factor, anisotropy_strength, anisotropy_cos, anisotropy_sin;
    vec3 sheen_color_factor; float sheen_roughness_factor;
    float transmission_factor; uint32_t bound_mask;
    const uint4* trio;                 // valid when bound_mask & RM_TRIO
};
PT_DEV MatHeader load_mat_header(const RMat* m) {
    const float4* q = (const float4*)m;
    const float4 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4], f = q[5], g = q[6], h = q[7];
    MatHeader r;
    r.flags = __float_as_uint(a.x); r.alpha_mode = __float_as_int(a.y); r.metalness_factor = a.z; r.roughness_factor = a.w;
    r.base_color_factor = {b.x, b.y, b.z, b.w};
    r.emissive_factor = v3(c.x, c.y, c.z); r.alpha_cutoff = c.w;
    r.ior = d.x; r.normal_scale = d.y; r.specular_factor = d.z; r.clearcoat_normal_scale = d.w;
    r.specular_color_factor = v3(e.x, e.y, e.z); r.clearcoat_factor = e.w;
    r.clearcoat_roughness_factor = f.x; r.anisotropy_strength = f.y; r.anisotropy_cos = f.z; r.anisotropy_sin = f.w;
    r.sheen_color_factor = v3(g.x, g.y, g.z); r.sheen_roughness_factor = g.w;
    r.transmission_factor = h.x; r.bound_mask = __float_as_uint(h.y);
    r.trio = (const uint4*)(((uint64_t)__float_as_uint(h.w) << 32) | (uint64_t)__float_as_uint(h.z));
    return r;
}
PT_DEV bool slot_bound(uint32_t mask, int slot) { return (mask >> slot) & 1u; }
// The shade stage gathers 17 dwordx4 of material data per hit (header + the albedo / normal / metal-rough slots) from a table
// of a few dozen records; that stage is bound by the issue rate of its vector-memory instructions, so a kernel built with
// PT_LUT_LDS keeps those 272 B of every material in LDS when the scene has few enough and reads them with ds_read instead.
#ifdef PT_LUT_LDS
constexpr uint32_t kMatCacheMax = 96;
constexpr uint32_t kMatCacheRecs = 17;                     // float4 per material: 8 header + 3 slots x 3
static __shared__ float4 pt_lds_mat[kMatCacheMax * kMatCacheRecs];
PT_DEV bool materials_cached(const SceneRec& sc) { return sc.small_tables || sc.n_materials <= kMatCacheMax; }
PT_DEV void stage_materials(const SceneRec& sc) {          // call once per workgroup, all threads
    if (materials_cached(sc)) {
        const uint32_t total = sc.n_materials * kMatCacheRecs;
        for (uint32_t i = threadIdx.x; i < total; i += blockDim.x) {
            const uint32_t m = i / kMatCacheRecs, k = i - m * kMatCacheRecs;
            pt_lds_mat[i] = ((const float4*)(sc.rmats + m))[k];
        }
    }
    __syncthreads();
}
PT_DEV MatHeader material_header(const SceneRec& sc, uint32_t mid) {
    if (materials_cached(sc)) return load_mat_header((const RMat*)(pt_lds_mat + mid * kMatCacheRecs));
    return load_mat_header(sc.rmats + mid);
}
PT_DEV RTex material_slot012(const SceneRec& sc, uint32_t mid, int slot) {       // slots 0..2 only (normal, albedo, metal-rough)
    if (materials_cached(sc)) return load_rtex((const RTex*)(pt_lds_mat + mid * kMatCacheRecs + 8 + 3 * slot));
    return load_rtex(&sc.rmats[mid].tex[slot]);
}
#else
PT_DEV MatHeader material_header(const SceneRec& sc, uint32_t mid) { return load_mat_header(sc.rmats + mid); }
PT_DEV RTex material_slot012(const SceneRec& sc, uint32_t mid, int slot) { return load_rtex(&sc.rmats[mid].tex[slot]); }
#endif
// GetBaseColor + GetAlpha for the any-hit paths (Material.hlsli:98-117); only the fields they need are loaded.
PT_DEV void base_color_alpha(const SceneRec& sc, const RMat* m, const vec2 tc[2], vec4 vc, unsigned& taps, float& base_alpha, float& alpha, float& cutoff) {
    const float4* q = (const float4*)m;
    const float4 a = q[0], b = q[1], c = q[2], h = q[7];
    vec4 col = vec4{b.x, b.y, b.z, b.w} * vc;
    if (slot_bound(__float_as_uint(h.y), SLOT_ALBEDO)) col = col * sample_slot(sc, m, SLOT_ALBEDO, tc, taps);
    const int alpha_mode = __float_as_int(a.y);
    cutoff = c.w;
    base_alpha = col.w;
    alpha = alpha_mode == PT_ALPHA_MODE_BLEND ? col.w : (alpha_mode == PT_ALPHA_MODE_MASK ? (col.w < cutoff ? 0.f : 1.f) : 1.f);
}
PT_DEV vec3 normal_from_sample(vec4 s, float scale, vec3 gn, vec3 t, vec3 b) {                        // Material.hlsli:119-128,199-208
    vec3 nm = v3(s.x * 2.f - 1.f, s.y * 2.f - 1.f, s.z * 2.f - 1.f);
    nm.x *= scale; nm.y *= scale;
    return normalize(to_world(t, b, gn, nm));
}
// `fetched`: the filtered emissive texel when it came with the interleaved footprint (RM_TRIO_EMISSIVE, get_surface), else ignored
PT_DEV vec3 emissive_of(const SceneRec& sc, const RMat* m, const MatHeader& h, const vec2 tc[2], unsigned& taps, vec3 fetched) {   // :151-159
    vec3 e = h.emissive_factor;
#if PT_TEX_PAIRS && PT_TEX_TRIO
    if (h.bound_mask & RM_TRIO_EMISSIVE) { taps++; return e * fetched; }
#endif
    if (slot_bound(h.bound_mask, SLOT_EMISSIVE)) e = e * xyz(sample_slot(sc, m, SLOT_EMISSIVE, tc, taps));
    return e;
}
PT_DEV vec3 normal_adaptation(vec3 ng, vec3 ns, vec3 v) {          // PathTracer.lib.hlsl:306-316
    vec3 r = reflect(-v, ns);
    float rdng = dot(r, ng);
    if (rdng < 0) return normalize(v + normalize(r - rdng * ng));
    return ns;
}
PT_DEV Surface get_surface(const SceneRec& sc, uint32_t flags, const RMat* m, const MatHeader& h, const HitGeom& a, vec3 view, unsigned& taps) {
    Surface s;
    s.emissive_texel = v3(0);
    // The three usual PBR textures are fetched as ONE batch: their slot records are loaded together, their twelve texel
    // gathers are issued together (unbound slots read a 1x1 white texel, so there is no branch to split the batch).
    const uint32_t mid = (uint32_t)(m - sc.rmats);
#if PT_TEX_PAIRS && PT_TEX_TRIO
    // Materials whose three textures share one footprint (RM_TRIO, the usual glTF PBR set) read it from the interleaved copy: one set
    // of addresses and weights, two dwordx4 per texel row.  Both paths are wave-uniform branches; a wave of trio materials only (every
    // wave of the bench scene) never enters the general one.
    const bool trio = (h.bound_mask & RM_TRIO) != 0;
    const RTex t_alb = material_slot012(sc, mid, SLOT_ALBEDO);
    const TexTaps k_alb = texture_taps(t_alb, a.tc);
    TexTaps k_nrm = k_alb, k_mr = k_alb;
    k_nrm.srgb = (h.bound_mask & RM_TRIO_SRGB_N) ? (uint32_t)RT_SRGB : 0u; k_mr.srgb = (h.bound_mask & RM_TRIO_SRGB_M) ? (uint32_t)RT_SRGB : 0u;
    uint32_t a00 = 0, a10 = 0, a01 = 0, a11 = 0, n00 = 0, n10 = 0, n01 = 0, n11 = 0, m00 = 0, m10 = 0, m01 = 0, m11 = 0;
    if (__any(trio)) {
        if (trio) {
            const uint4* row0 = h.trio + (k_alb.p0 - t_alb.texels) + k_alb.ia;
            const uint4* row1 = h.trio + (k_alb.p1 - t_alb.texels) + k_alb.ia;
            const uint4 A0 = gload_u4(row0), B0 = gload_u4(row0 + 1), A1 = gload_u4(row1), B1 = gload_u4(row1 + 1);
            uint4 t00 = k_alb.i0 == k_alb.ia ? A0 : B0, t01 = k_alb.i0 == k_alb.ia ? A1 : B1;
            uint4 t10 = k_alb.i1 == k_alb.ia ? A0 : B0, t11 = k_alb.i1 == k_alb.ia ? A1 : B1;
            if (__any(k_alb.edge)) {
                if (k_alb.edge) { t10 = gload_u4(h.trio + (k_alb.p0 - t_alb.texels) + k_alb.i1); t11 = gload_u4(h.trio + (k_alb.p1 - t_alb.texels) + k_alb.i1); }
            }
            a00 = t00.x; a10 = t10.x; a01 = t01.x; a11 = t11.x;
            n00 = t00.y; n10 = t10.y; n01 = t01.y; n11 = t11.y;
            m00 = t00.z; m10 = t10.z; m01 = t01.z; m11 = t11.z;
            if (h.bound_mask & RM_TRIO_EMISSIVE) {
                TexTaps k_em = k_alb;
                k_em.srgb = (h.bound_mask & RM_TRIO_SRGB_E) ? (uint32_t)RT_SRGB : 0u;
                s.emissive_texel = xyz(resolve_taps(k_em, t00.w, t10.w, t01.w, t11.w, sc.srgb_lut));
            }
        }
    }
    if (__any(!trio)) {
        if (!trio) {
            const RTex t_nrm = material_slot012(sc, mid, SLOT_NORMAL), t_mr = material_slot012(sc, mid, SLOT_METALLIC_ROUGHNESS);
            k_nrm = texture_taps(t_nrm, a.tc); k_mr = texture_taps(t_mr, a.tc);
            const uint2 ar0 = tap_row(k_alb, 0), ar1 = tap_row(k_alb, 1), nr0 = tap_row(k_nrm, 0), nr1 = tap_row(k_nrm, 1), mr0 = tap_row(k_mr, 0), mr1 = tap_row(k_mr, 1);
            const TexQuad qa = tap_quad(k_alb, ar0, ar1), qn = tap_quad(k_nrm, nr0, nr1), qm = tap_quad(k_mr, mr0, mr1);
            a00 = qa.t00; a10 = qa.t10; a01 = qa.t01; a11 = qa.t11;
            n00 = qn.t00; n10 = qn.t10; n01 = qn.t01; n11 = qn.t11;
            m00 = qm.t00; m10 = qm.t10; m01 = qm.t01; m11 = qm.t11;
        }
    }
#else
    const RTex t_alb = material_slot012(sc, mid, SLOT_ALBEDO), t_nrm = material_slot012(sc, mid, SLOT_NORMAL), t_mr = material_slot012(sc, mid, SLOT_METALLIC_ROUGHNESS);
    const TexTaps k_alb = texture_taps(t_alb, a.tc), k_nrm = texture_taps(t_nrm, a.tc), k_mr = texture_taps(t_mr, a.tc);
#if PT_TEX_PAIRS
    const uint2 ar0 = tap_row(k_alb, 0), ar1 = tap_row(k_alb, 1), nr0 = tap_row(k_nrm, 0), nr1 = tap_row(k_nrm, 1), mr0 = tap_row(k_mr, 0), mr1 = tap_row(k_mr, 1);
    const TexQuad qa = tap_quad(k_alb, ar0, ar1), qn = tap_quad(k_nrm, nr0, nr1), qm = tap_quad(k_mr, mr0, mr1);
    const uint32_t a00 = qa.t00, a10 = qa.t10, a01 = qa.t01, a11 = qa.t11;
    const uint32_t n00 = qn.t00, n10 = qn.t10, n01 = qn.t01, n11 = qn.t11;
    const uint32_t m00 = qm.t00, m10 = qm.t10, m01 = qm.t01, m11 = qm.t11;
#else
    const uint32_t a00 = gload(k_alb.p00), a10 = gload(k_alb.p10), a01 = gload(k_alb.p01), a11 = gload(k_alb.p11);
    const uint32_t n00 = gload(k_nrm.p00), n10 = gload(k_nrm.p10), n01 = gload(k_nrm.p01), n11 = gload(k_nrm.p11);
    const uint32_t m00 = gload(k_mr.p00), m10 = gload(k_mr.p10), m01 = gload(k_mr.p01), m11 = gload(k_mr.p11);
#endif
#endif
    const bool b_alb = slot_bound(h.bound_mask, SLOT_ALBEDO), b_nrm = slot_bound(h.bound_mask, SLOT_NORMAL), b_mr = slot_bound(h.bound_mask, SLOT_METALLIC_ROUGHNESS);
    taps += (b_alb ? 1u : 0u) + (b_nrm ? 1u : 0u) + (b_mr ? 1u : 0u);
    vec4 bc = h.base_color_factor * a.color;                                                      // GetBaseColor, Material.hlsli:98-106
    if (b_alb) bc = bc * resolve_taps(k_alb, a00, a10, a01, a11, sc.srgb_lut);
    s.albedo = xyz(bc);
    s.alpha = h.alpha_mode == PT_ALPHA_MODE_BLEND ? bc.w : (h.alpha_mode == PT_ALPHA_MODE_MASK ? (bc.w < h.alpha_cutoff ? 0.f : 1.f) : 1.f);   // :108-117
    s.n = b_nrm ? normal_from_sample(resolve_taps(k_nrm, n00, n10, n01, n11, sc.srgb_lut), h.normal_scale, a.n, a.t, a.bt) : a.n;
    if (flags & PT_FLAG_SHADING_NORMAL_ADAPTATION) s.n = normal_adaptation(a.ng, s.n, view);
    float metal = h.metalness_factor, rough = h.roughness_factor;                                 // Material.hlsli:130-140
    if (b_mr) { vec4 t = resolve_taps(k_mr, m00, m10, m01, m11, sc.srgb_lut); metal *= t.z; rough *= t.y; }
    s.metalness = metal;
    s.ay = hmax(rough * rough, kMinRoughness);
    // PathTracer.lib.hlsl:339,341: occlusion and the first emissive fetch are dead values; the fetches are
    // skipped here (no visible effect) but still counted so tap counters match the reference's traffic.
    if (slot_bound(h.bound_mask, SLOT_OCCLUSION)) taps++;
    if (slot_bound(h.bound_mask, SLOT_EMISSIVE)) taps++;
    s.ior = h.ior;
    s.spec_factor = h.specular_factor;                                                            // :161-168
    s.spec_color = h.specular_color_factor;                                                       // :170-177
    s.clearcoat = h.clearcoat_factor;                                                             // :179-186
    s.cc_rough = h.clearcoat_roughness_factor;                                                    // :188-195
    s.cc_n = a.n;
    float strength = h.anisotropy_strength;
    vec3 av = v3(1, 0, 1);
    s.sheen_color = h.sheen_color_factor;                                                         // :210-217
    float sheen_rough = h.sheen_roughness_factor;                                                 // :219-226
    s.transmissive = h.transmission_factor;                                                       // :228-235
#ifndef PT_PROBE_BASE_ONLY
    if ((h.bound_mask & ((1u << SLOT_COUNT) - 1u)) >> SLOT_SPECULAR) {   // any of the rarely-bound extension textures (slots 5..14)
        if (slot_bound(h.bound_mask, SLOT_SPECULAR)) s.spec_factor *= sample_slot(sc, m, SLOT_SPECULAR, a.tc, taps).w;
        if (slot_bound(h.bound_mask, SLOT_SPECULAR_COLOR)) s.spec_color = s.spec_color * xyz(sample_slot(sc, m, SLOT_SPECULAR_COLOR, a.tc, taps));
        if (slot_bound(h.bound_mask, SLOT_CLEARCOAT)) s.clearcoat *= sample_slot(sc, m, SLOT_CLEARCOAT, a.tc, taps).x;
        if (slot_bound(h.bound_mask, SLOT_CLEARCOAT_ROUGHNESS)) s.cc_rough *= sample_slot(sc, m, SLOT_CLEARCOAT_ROUGHNESS, a.tc, taps).y;
        if (slot_bound(h.bound_mask, SLOT_CLEARCOAT_NORMAL))
            s.cc_n = normal_from_sample(sample_slot(sc, m, SLOT_CLEARCOAT_NORMAL, a.tc, taps), h.clearcoat_normal_scale, a.n, a.t, a.bt);
        if (slot_bound(h.bound_mask, SLOT_ANISOTROPY)) {   // GetAnisotropyStrengthAndDirection (Material.hlsli:246-262)
            vec4 t = sample_slot(sc, m, SLOT_ANISOTROPY, a.tc, taps);
            av = v3(t.x * 2 - 1, t.y * 2 - 1, t.z);
        }
        if (slot_bound(h.bound_mask, SLOT_SHEEN_COLOR)) s.sheen_color = s.sheen_color * xyz(sample_slot(sc, m, SLOT_SHEEN_COLOR, a.tc, taps));
        if (slot_bound(h.bound_mask, SLOT_SHEEN_ROUGHNESS)) sheen_rough *= sample_slot(sc, m, SLOT_SHEEN_ROUGHNESS, a.tc, taps).w;
        if (slot_bound(h.bound_mask, SLOT_TRANSMISSION)) s.transmissive *= sample_slot(sc, m, SLOT_TRANSMISSION, a.tc, taps).x;
        if (slot_bound(h.bound_mask, SLOT_THICKNESS)) taps++;   // thickness is loaded upstream but unused (quirk q14)
    }
#endif
    if (flags & PT_FLAG_SHADING_NORMAL_ADAPTATION) s.cc_n = normal_adaptation(a.ng, s.cc_n, view);
    const float cr = h.anisotropy_cos, sr = h.anisotropy_sin;
    vec2 adir = normalize(vec2{cr * av.x + -sr * av.y, sr * av.x + cr * av.y});
    strength *= av.z;
    // CalculateShadingTangentAndBitangent (Material.hlsli:264-270)
    vec3 sb = normalize(cross(s.n, a.t));
    vec3 st = normalize(cross(sb, s.n));
    sb = sb * a.tw;
    s.at = normalize(to_world(st, sb, s.n, v3(adir.x, adir.y, 0)));
    s.ab = normalize(cross(s.at, s.n));
    s.ax = hmax(lerpf(s.ay, 1, strength * strength), kMinRoughness);
    s.sheen_a = hmax(sheen_rough * sheen_rough, kMinRoughness);
    // ClosestHit :856-861
    s.ax = hmax(s.ax, kMinRoughness); s.ay = hmax(s.ay, kMinRoughness);
    s.cc_rough = hmax(s.cc_rough, kMinRoughness);
    if (flags & PT_FLAG_MATERIAL_USE_GEOMETRIC_NORMALS) { s.n = a.ng; s.cc_n = a.ng; }
    return s;
}

// ---------------------------------------------------------------- BSDF terms (Bsdf.hlsli)
PT_DEV float schlick(float f0, float c) { return f0 + (1 - f0) * hpow5(1 - fabsf(c)); }                    // :39-42
PT_DEV vec3 schlick3(vec3 f0, float c) { return f0 + (1 - f0) * hpow5(1 - fabsf(c)); }                     // :44-47
PT_DEV float ggx_d(float a, float ndh) {                                                                       // :50-57
    float a2 = a * a;
    float den = ndh * ndh * (a2 - 1) + 1;
    den *= kPi * den;
    return fdiv(a2 * heavyside(ndh), den);
}
PT_DEV float ggx_corr_v(float a, float ndl, float ndv, float hdl, float hdv) {                                 // :78-85
    float a2 = a * a;
    float num = 0.5f * heavyside(hdl) * heavyside(hdv);
    float den = fabsf(ndv) * sqrtf(a2 + (1 - a2) * ndl * ndl);
    den += fabsf(ndl) * sqrtf(a2 + (1 - a2) * ndv * ndv);
    return fdiv(num, den);
}
PT_DEV float specular_brdf(float a, float ndl, float ndv, float ndh, float hdl, float hdv) { return ggx_corr_v(a, ndl, ndv, hdl, hdv) * ggx_d(a, ndh); }  // :87-90
PT_DEV float ggx_aniso_d(float ax, float ay, vec3 h) {                                                         // :93-99
    float a2 = ax * ay;
    vec3 f = v3(ay * h.x, ax * h.y, a2 * h.z);
    float w2 = fdiv(a2, dot(f, f));
    return fdiv(heavyside(h.z) * a2 * w2 * w2, kPi);
}
PT_DEV float aniso_specular_brdf(float ax, float ay, vec3 v, vec3 h, vec3 l) {                                 // :117-130
    float hdv = dot(h, v), hdl = dot(h, l);
    float num = 0.5f * heavyside(hdv) * heavyside(hdl);
    float vv = fabsf(l.z) * length(v3(ax * v.x, ay * v.y, v.z));
    float ll = fabsf(v.z) * length(v3(ax * l.x, ay * l.y, l.z));
    return fdiv(num, vv + ll) * ggx_aniso_d(ax, ay, h);
}
PT_DEV float fresnel_coat_w(float weight, float ndv) {            // FresnelCoat's lerp factor, IOR 1.5 (:157-163)
    float f0 = (1 - 1.5f) / (1 + 1.5f);
    f0 *= f0;
    return weight * schlick(f0, ndv);
}
PT_DEV float sheen_l(float alpha, float x) {                                                                   // :175-184
    float t = (1 - alpha) * (1 - alpha);
    float a = lerpf(21.5473f, 25.3245f, t), b = lerpf(3.82987f, 3.32435f, t), c = lerpf(0.19823f, 0.16801f, t);
    float d = lerpf(-1.97760f, -1.27393f, t), e = lerpf(-4.32054f, -4.85967f, t);
    return fdiv(a, 1 + b * hpow(x, c)) + d * x + e;
}
// `l_half` = sheen_l(alpha, 0.5f)
PT_DEV float sheen_shadowing(float alpha, float c, float l_half) {                                             // :186-193
    if (c < 0.5f) return pt_exp(sheen_l(alpha, c));
    return pt_exp(2 * l_half - sheen_l(alpha, 1 - c));
}
PT_DEV float sheen_alpha(const Surface& s) { return clampf(s.sheen_a, 0.000001f, 1); }
// The parts of SheenBrdf that do not depend on the light direction, once per vertex instead of once per evaluation (three a hit: the two
// light samples and the sampled direction): SheenL(alpha, 1/2) and the view direction's shadowing term -- 2 to 4 of the lobe's 5 to 7 pows
// and one of its two exps.  Pure functions of the same arguments: the same bits as evaluating them in place.
PT_DEV void prepare_sheen(Surface& s, vec3 v) {
    const float sa = sheen_alpha(s);
    s.sheen_lh = sheen_l(sa, 0.5f);
    s.sheen_sv = sheen_shadowing(sa, to_local(s.at, s.ab, s.n, v).z, s.sheen_lh);
}
PT_DEV float sheen_brdf(const Surface& s, float alpha, float ndl, float ndv, float ndh) {                      // :166-173,195-203
    float inv_r = fdiv(1.0f, alpha);
    float sin2h = 1 - ndh * ndh;
    float d = fdiv((2 + inv_r) * hpow(sin2h, inv_r * 0.5f), 2 * kPi);
    // SheenBrdf passes (n_dot_v, n_dot_l) into SheenVisibility(alpha, n_dot_l, n_dot_v): swapped names, same product
    float vis = clampf(fdiv(1.0f, (1 + s.sheen_sv + sheen_shadowing(alpha, ndl, s.sheen_lh)) * 4 * ndv * ndl), 0, 1);
    return d * vis;
}
PT_DEV float sheen_e(const float* lut, float alpha, float cos_theta) {   // Bsdf.hlsli:204-208: bilinear, clamp, 16x16
    float x = cos_theta * 16.f - 0.5f, y = alpha * 16.f - 0.5f;
    if (!(x == x)) x = 0;
    if (!(y == y)) y = 0;
    x = clampf(x, -1.f, 16.f); y = clampf(y, -1.f, 16.f);
    float fx0 = floorf(x), fy0 = floorf(y);
    int i0 = (int)fx0, j0 = (int)fy0;
    float fx = x - fx0, fy = y - fy0;
    int ia = max(i0, 0), ib = min(i0 + 1, 15), ja = max(j0, 0), jb = min(j0 + 1, 15);
    ia = min(ia, 15); ja = min(ja, 15); ib = max(ib, 0); jb = max(jb, 0);
    float w00 = (1 - fx) * (1 - fy), w10 = fx * (1 - fy), w01 = (1 - fx) * fy, w11 = fx * fy;
    return sheen_entry(lut, ja * 16 + ia) * w00 + sheen_entry(lut, ja * 16 + ib) * w10 + sheen_entry(lut, jb * 16 + ia) * w01 + sheen_entry(lut, jb * 16 + ib) * w11;
}
PT_DEV float modulate_roughness(float a, float ior) { return clampf(lerpf(0, a, saturate(2 * (ior - 1))), kMinRoughness, 1.0f); }   // :216-220

// GltfBsdf, both overloads (Bsdf.hlsli:241-325).  mode 0: overload without the flag (all terms);
// mode 1: is_transmission == false; mode 2: is_transmission == true.
PT_DEV vec3 gltf_bsdf(const float* lut, const Surface& s, vec3 v, vec3 l, int mode) {
    vec3 n = s.n;
    vec3 h = normalize(v + l);
    vec3 vl = to_local(s.at, s.ab, n, v), hl = to_local(s.at, s.ab, n, h), ll = to_local(s.at, s.ab, n, l);
    float hdl = dot(h, l), hdv = dot(h, v);
    float h_dot_abs_l = dot(normalize(v3(ll.x, ll.y, fabsf(ll.z)) + vl), vl);
    bool refl = mode != 2, trans = mode != 1;
    // the coat first (PT_COAT_FIRST; it is the LAST term of the shader's expression, :318-324): its inputs are the world-space n, v, h, l, which
    // nothing after it needs -- evaluated at the end they stayed live through the whole function in a kernel that has no register to spare
    float cndv = dot(n, v), cndh = dot(n, h), cndl = dot(n, l);                    // (sic) shading normal
#ifndef PT_PROBE_BASE_ONLY
    const float cc = refl ? saturate(cndl) * specular_brdf(s.cc_rough, cndl, cndv, cndh, hdl, hdv) : 0.f;
#else
    const float cc = 0.f;
#endif
    const float coat_w = fresnel_coat_w(s.clearcoat, cndv);
    float spec = refl ? saturate(ll.z) * aniso_specular_brdf(s.ax, s.ay, vl, hl, ll) : 0.f;
    vec3 diffuse = refl ? saturate(ll.z) * (s.albedo / kPi) : v3(0);
    vec3 transmission = v3(0);
#ifndef PT_PROBE_BASE_ONLY
    if (trans) {                                           // ThinSurfaceTransmissionBtdf :222-228
        float a = modulate_roughness(s.ay, s.ior);
        vec3 lr = l - 2 * dot(n, l) * n;
        vec3 hr = normalize(v + lr);
        transmission = saturate(-ll.z) * (s.albedo * specular_brdf(a, dot(n, lr), dot(n, v), dot(n, hr), dot(hr, lr), dot(hr, v)));
    }
#endif
    diffuse = lerp3(diffuse, transmission, s.transmissive);
    // FresnelMix :137-144
    float f0s = fdiv(1 - s.ior, 1 + s.ior);
    vec3 f0 = v3(f0s);
    f0 *= f0 * s.spec_color;
    f0 = hmin(f0, v3(1));
    vec3 fr = schlick3(f0, h_dot_abs_l);
    vec3 dielectric = (1 - s.spec_factor * max3(fr)) * diffuse + s.spec_factor * fr * v3(spec);
    vec3 metal = refl ? v3(spec) * schlick3(s.albedo, hdv) : v3(0);                // ConductorFresnel :146-149
    vec3 material = lerp3(dielectric, metal, s.metalness);
    float sa = sheen_alpha(s);
    float ms = max3(s.sheen_color);                                               // SheenMix :210-214
#ifndef PT_SHEEN_SKIP
#define PT_SHEEN_SKIP 1
#endif
    // A material WITHOUT sheen (sheen colour exactly (0,0,0): every material that does not use KHR_materials_sheen) in a wave that holds no
    // sheen material skips the sheen BRDF -- two pow, three SheenL, the exponentials: ~150 instructions, three evaluations a hit -- and gets
    // the SAME bits: the layer term is 0 * sheen_brdf, which is +0 whenever sheen_brdf is finite and NaN otherwise, and sheen_brdf =
    // saturate(.) * D * V with V clamped to [0, 1] (a NaN clamps to 0) and D = (2 + 1/a) pow(1 - ndh^2, 1/(2a)) / 2 pi, finite unless the pow's
    // base is negative or NaN (|ndh| a rounding above 1, a NaN normal); the albedo scaling is 1 - 0 * E = 1 exactly.
    const bool no_sheen = s.sheen_color.x == 0.0f && s.sheen_color.y == 0.0f && s.sheen_color.z == 0.0f;
    if (PT_SHEEN_SKIP && !__any(!no_sheen)) {
        const float sin2h = 1 - hl.z * hl.z;
        const float layer = (refl && !(sin2h >= 0.0f)) ? __builtin_nanf("") : 0.0f;
        material = v3(layer) + material * 1.0f;
    } else {
#ifndef PT_PROBE_BASE_ONLY
        vec3 sheen = refl ? v3(saturate(ll.z) * sheen_brdf(s, sa, ll.z, vl.z, hl.z)) : v3(0);
#else      // PROBE ONLY (tools/build_variant.sh): what the extension lobes cost a hit that has none; wrong for materials that do
        vec3 sheen = v3(0);
#endif
        // Without sheen (ms == 0) both terms are 1 - 0 * E = 1 exactly (E is always finite: sheen_e clamps NaN coordinates):
        // skip the eight table gathers.
        float scaling = 1.0f;
        if (ms != 0.0f) scaling = hmin(1.0f - ms * sheen_e(lut, sa, vl.z), 1.0f - ms * sheen_e(lut, sa, ll.z));
        material = s.sheen_color * sheen + material * scaling;
    }
    return lerp3(material, v3(cc), coat_w);                                         // FresnelCoat(1.5, ...)
}

// ---------------------------------------------------------------- sampling (Sampling.hlsli, Transforms.hlsli)
PT_DEV vec2 uv_to_square(vec2 uv) { return {uv.x * 2 + -1, uv.y * -2 + 1}; }                                   // Transforms.hlsli:52-55
PT_DEV vec2 square_to_uv(vec2 s) { return {(s.x - -1) * 0.5f, (s.y - 1) * -0.5f}; }                            // :57-60
// Each direction sampler is split into the angle it takes the sine and cosine of and the rest: sample_bsdf's lobes diverge inside a wave
// (half the lanes sample the cosine lobe, half the specular one), and ONE sinf / cosf pair of a per-lane angle serves every lobe -- the two
// ~125-instruction calls used to run once per lobe present in the wave, each with part of the lanes.  Same arguments, same bits.
PT_DEV float square_to_disk_angle(vec2 s, float& r) {
    r = hmax(fabsf(s.x), fabsf(s.y));
    return r == 0 ? 0 : fdiv(kPi * (r + (fabsf(s.y) - fabsf(s.x))), 4 * r);
}
PT_DEV vec2 square_to_disk_finish(vec2 s, float r, float cs, float sn) { return {signf(s.x) * r * cs, signf(s.y) * r * sn}; }
PT_DEV vec2 square_to_disk(vec2 s) {                                                                           // :83-90
    float r;
    const float phi = square_to_disk_angle(s, r);
    float sn_, cs_;
    pt_sincos(phi, sn_, cs_);
    return square_to_disk_finish(s, r, cs_, sn_);
}
PT_DEV vec3 square_to_sphere(vec2 s) {                                                                         // :124-136
    float d = 1 - (fabsf(s.x) + fabsf(s.y));
    float r = 1 - fabsf(d);
    float phi = (r == 0) ? 0 : (kPi / 4) * (fdiv(fabsf(s.y) - fabsf(s.x), r) + 1);
    float f = r * sqrtf(2 - r * r);
    float sn_, cs_;
    pt_sincos(phi, sn_, cs_);
    return {f * signf(s.x) * cs_, f * signf(s.y) * sn_, signf(d) * (1 - r * r)};
}
PT_DEV vec2 sphere_to_square(vec3 p) {                                                                         // :138-149
    float r = sqrtf(1 - fabsf(p.z));
    float phi = pt_atan2(fabsf(p.y), fabsf(p.x));
    float d = signf(p.z) * (1 - r);
    float diff = r * ((4 / kPi) * phi - 1);
    return {signf(p.x) * 0.5f * (1 - d - diff), signf(p.y) * 0.5f * (1 - d + diff)};
}
PT_DEV vec3 cubemap_to_direction(int face, float u, float v) {                                                 // :10-50
    vec3 ud, vd, fd;
    switch (face) {
        case 0: fd = v3(1, 0, 0); ud = v3(0, 0, -1); vd = v3(0, -1, 0); break;
        case 1: fd = v3(-1, 0, 0); ud = v3(0, 0, 1); vd = v3(0, -1, 0); break;
        case 2: fd = v3(0, 1, 0); ud = v3(1, 0, 0); vd = v3(0, 0, 1); break;
        case 3: fd = v3(0, -1, 0); ud = v3(1, 0, 0); vd = v3(0, 0, -1); break;
        case 4: fd = v3(0, 0, 1); ud = v3(1, 0, 0); vd = v3(0, -1, 0); break;
        default: fd = v3(0, 0, -1); ud = v3(-1, 0, 0); vd = v3(0, -1, 0); break;
    }
    u = u * 2 - 1; v = v * 2 - 1;
    return normalize(fd + u * ud + v * vd);
}
PT_DEV vec3 sample_cosine_hemisphere_finish(vec3 n, float u1, float cs, float sn) {       // cs, sn = cos, sin of kTau * u0
    float y = 2 * u1 - 1;
    float s = sqrtf(1.0f - y * y);
    return normalize(n + v3(s * cs, s * sn, y));
}
PT_DEV vec3 sample_cosine_hemisphere(vec3 n, float u0, float u1) {                                             // Sampling.hlsli:26-33
    float theta = kTau * u0;
    float sn_, cs_;
    pt_sincos(theta, sn_, cs_);
    return sample_cosine_hemisphere_finish(n, u1, cs_, sn_);
}
PT_DEV float cosine_hemisphere_pdf(vec3 n, vec3 v) { return saturate(fdiv(dot(v, n), kPi)); }                       // :35-38
PT_DEV vec3 sample_ggx_normal_finish(float a, float u1, float cs, float sn) {              // cs, sn = cos, sin of kTau * u0
    float ct = sqrtf(fdiv(1 - u1, 1 + (a * a - 1) * u1));
    float st = sqrtf(1 - ct * ct);
    return {st * cs, st * sn, ct};
}
PT_DEV vec3 sample_ggx_normal(float a, float u0, float u1) {                                                   // :41-52
    float phi = kTau * u0;
    float sn_, cs_;
    pt_sincos(phi, sn_, cs_);
    return sample_ggx_normal_finish(a, u1, cs_, sn_);
}
PT_DEV float ggx_normal_pdf(float a, vec3 n, vec3 h) { float ndh = dot(n, h); return ggx_d(a, ndh) * ndh; }    // :54-58

// ---------------------------------------------------------------- lobe logic (PathTracer.lib.hlsl:383-667)
struct Lobes { float alpha, clearcoat, sheen, specular, diffuse, transmission; };
PT_DEV Lobes lobe_probabilities(const Surface& s, vec3 v) {                                                   // :535-553
    Lobes p;
    float remaining = 1;
    p.alpha = 1.0f - s.alpha;
    remaining -= p.alpha;
    p.clearcoat = lerpf(0.f, 1.f, fresnel_coat_w(s.clearcoat, dot(s.cc_n, v)));
    p.clearcoat *= remaining;
    remaining -= p.clearcoat;
    p.sheen = any_gt0(s.sheen_color) ? 0.5f : 0.0f;
    p.sheen *= remaining;
    remaining -= p.sheen;
    p.specular = 0.5f;
    p.specular *= remaining;
    remaining -= p.specular;
    p.transmission = s.transmissive;
    p.transmission *= remaining;
    remaining -= p.transmission;
    p.diffuse = remaining;
    return p;
}
PT_DEV float transmission_pdf(const Surface& s, vec3 v, vec3 l) {                                             // :489-500
    float a = modulate_roughness(s.ay, s.ior);
    l = l - 2 * dot(s.n, l) * s.n;
    vec3 h = normalize(v + l);
    float pdf = ggx_normal_pdf(a, s.n, h);
    pdf = fdiv(pdf, 4 * dot(v, h));
    return pdf;
}
PT_DEV float bsdf_pdf(const Surface& s, vec3 v, vec3 l, bool is_transmission, const Lobes& p) {               // :555-565
    if (is_transmission) return p.transmission * transmission_pdf(s, v, l);
    vec3 h = normalize(v + l);
    float vdh4 = 4 * dot(v, h);
    float cc = ggx_normal_pdf(s.cc_rough, s.cc_n, h);                                                        // ClearcoatPdf :408-416
    cc = fdiv(cc, vdh4);
    float pdf = p.clearcoat * cc;
    float cosp = cosine_hemisphere_pdf(s.n, l);
    pdf += p.sheen * cosp;                                                                                   // SheenPdf :423-426
    vec3 lh = to_local(s.at, s.ab, s.n, h);                                                                  // SpecularPdf :444-460
    float sp = ggx_aniso_d(s.ax, s.ay, lh) * lh.z;
    sp = fdiv(sp, vdh4);
    pdf += p.specular * sp;
    pdf += p.diffuse * cosp;                                                                                 // DiffusePdf :467-470
    return pdf;
}
PT_DEV vec3 evaluate_bsdf(uint32_t flags, const float* lut, const Surface& s, const Lobes& p, vec3 ng, vec3 v, vec3 l, float& pdf) {   // :567-593
    if (flags & PT_FLAG_MATERIAL_DIFFUSE_WHITE) {
        float ndl = saturate(dot(s.n, l));
        pdf = fdiv(ndl, kPi);
        return v3(fdiv(ndl, kPi));
    }
    if (flags & PT_FLAG_MATERIAL_MIS) {
        bool is_transmission = (dot(ng, l) * dot(ng, v)) < 0;
        pdf = bsdf_pdf(s, v, l, is_transmission, p);
        return s.alpha * gltf_bsdf(lut, s, v, l, is_transmission ? 2 : 1);
    }
    float ndl = saturate(dot(s.n, l));
    pdf = fdiv(ndl, kPi);
    pdf *= s.alpha;
    return s.alpha * gltf_bsdf(lut, s, v, l, 0);
}
PT_DEV vec3 sample_bsdf(uint32_t flags, const float* lut, const Surface& s, const Lobes& p, vec3 u, vec3 v, vec3& l, float& pdf,
                        bool& is_transmission, bool& use_mis) {                                             // :595-667
    if (flags & PT_FLAG_MATERIAL_DIFFUSE_WHITE) {
        use_mis = true; is_transmission = false;
        l = sample_cosine_hemisphere(s.n, u.y, u.z);
        pdf = cosine_hemisphere_pdf(s.n, l);
        return v3(fdiv(dot(s.n, l), kPi));
    }
    if (flags & PT_FLAG_MATERIAL_MIS) {
        is_transmission = false; use_mis = true;
        // SelectBsdf :511-533
        float x = u.x;
        int layer;
        if (x <= p.alpha) layer = 4;
        else { x -= p.alpha;
            if (x <= p.clearcoat) layer = 3;
            else { x -= p.clearcoat;
                if (x <= p.sheen) layer = 2;
                else { x -= p.sheen;
                    if (x <= p.specular) layer = 1;
                    else { x -= p.specular; layer = (x <= p.transmission) ? 5 : 0; } } } }
        if (layer == 4) {                                   // BSDF_LAYER_ALPHA
            l = -v; use_mis = false; pdf = p.alpha; is_transmission = true;
            return v3(1 - s.alpha);
        }
        // the angle each lobe's sampler takes the sine and cosine of, then one sinf / cosf pair for the whole wave (see square_to_disk_angle)
        const vec2 sq = uv_to_square({u.y, u.z});
        float disk_r = 0;
        const float disk_phi = square_to_disk_angle(sq, disk_r);
        const float angle = layer == 1 ? disk_phi : kTau * u.y;
        float cs, sn;
        pt_sincos(angle, sn, cs);
        if (layer == 0 || layer == 2) l = sample_cosine_hemisphere_finish(s.n, u.z, cs, sn);  // diffuse :462-465, sheen :418-421
        else if (layer == 1) {                              // SampleSpecular :428-442, SampleGgxAnisotropicNormal Sampling.hlsli:60-65
            vec2 d = square_to_disk_finish(sq, disk_r, cs, sn);
            vec3 hl = v3(d.x, d.y, sqrtf(1 - d.x * d.x - d.y * d.y));
            hl.x *= s.ax; hl.y *= s.ay;
            hl = normalize(hl);
            l = reflect(-v, to_world(s.at, s.ab, s.n, hl));
        } else if (layer == 3) {                            // SampleClearcoat :394-406
            vec3 t, b;
            basis_simple(s.cc_n, t, b);
            l = reflect(-v, to_world(t, b, s.cc_n, sample_ggx_normal_finish(s.cc_rough, u.z, cs, sn)));
        } else {                                            // SampleTransmission :472-487
            float a = modulate_roughness(s.ay, s.ior);
            vec3 h = to_world(s.at, s.ab, s.n, sample_ggx_normal_finish(a, u.z, cs, sn));
            l = reflect(-v, h);
            l = l - 2 * dot(s.n, l) * s.n;
            is_transmission = true;
        }
        pdf = bsdf_pdf(s, v, l, is_transmission, p);
        return s.alpha * gltf_bsdf(lut, s, v, l, is_transmission ? 2 : 1);
    }
    if (u.x > s.alpha) {
        l = -v; use_mis = false; pdf = (1 - s.alpha); is_transmission = true;
        return v3(1 - s.alpha);
    }
    use_mis = true; is_transmission = false;
    l = sample_cosine_hemisphere(s.n, u.y, u.z);
    pdf = cosine_hemisphere_pdf(s.n, l);
    pdf *= s.alpha;
    return s.alpha * gltf_bsdf(lut, s, v, l, 0);
}

// ---------------------------------------------------------------- lights (Lights.hlsli:26-61)
// The light table is tiny (64 B a light): the shade stage keeps up to kLightCacheMax lights in LDS, so picking one by a per-lane
// random index is a ds_read instead of a dependent global gather.
#ifdef PT_LUT_LDS
constexpr int kLightCacheMax = 32;
static __shared__ float4 pt_lds_light[kLightCacheMax * 4];
// A spot light's cone terms (Lights.hlsli:52-55: two cosines and a division that depend on the light alone) are evaluated ONCE per
// staged light, by the expressions light_ray() uses, and kept in the record's two padding floats of the LDS copy: every wave holds a
// lane that picked the spot light, so every hit used to pay for them.
PT_DEV void spot_cone_terms(float inner_angle, float outer_angle, float& scale, float& offset) {
    scale = fdiv(1.0f, hmax(0.001f, pt_cos(inner_angle) - pt_cos(outer_angle)));
    offset = -pt_cos(outer_angle) * scale;
}
PT_DEV void stage_lights(const SceneRec& sc, int num_of_lights) {   // 256-thread workgroups
    const uint32_t n = (uint32_t)(num_of_lights < kLightCacheMax ? num_of_lights : kLightCacheMax), n4 = n * 4u;
    if (threadIdx.x < n4) pt_lds_light[threadIdx.x] = gload_f4((const float4*)sc.lights + threadIdx.x);
    __syncthreads();
    if (threadIdx.x < n) {
        // pt_light as floats: [0] type [1-3] position [4] cutoff [5-7] direction [8] intensity [9-11] color [12] inner [13] outer [14-15] pad
        float4& last = pt_lds_light[threadIdx.x * 4u + 3u];        // inner_angle, outer_angle, pad, pad
        float scale, offset;
        spot_cone_terms(last.x, last.y, scale, offset);
        last.z = scale; last.w = offset;
    }
    __syncthreads();
}
PT_DEV pt_light load_light(const SceneRec& sc, uint32_t li, bool& cone_terms_staged) {
    float4 q[4];
    cone_terms_staged = sc.small_tables || li < (uint32_t)kLightCacheMax;
    if (cone_terms_staged) { const float4* p = pt_lds_light + li * 4u; q[0] = p[0]; q[1] = p[1]; q[2] = p[2]; q[3] = p[3]; }
    else { const float4* p = (const float4*)sc.lights + (size_t)li * 4u; q[0] = gload_f4(p); q[1] = gload_f4(p + 1); q[2] = gload_f4(p + 2); q[3] = gload_f4(p + 3); }
    pt_light l;
    static_assert(sizeof(pt_light) == 64, "pt_light");
    memcpy(&l, q, 64);
    return l;
}
#else
PT_DEV void spot_cone_terms(float inner_angle, float outer_angle, float& scale, float& offset) {
    scale = fdiv(1.0f, hmax(0.001f, pt_cos(inner_angle) - pt_cos(outer_angle)));
    offset = -pt_cos(outer_angle) * scale;
}
PT_DEV void stage_lights(const SceneRec&, int) {}
PT_DEV pt_light load_light(const SceneRec& sc, uint32_t li, bool& cone_terms_staged) { cone_terms_staged = false; return sc.lights[li]; }
#endif
// cone_terms_staged: the record's padding holds spot_cone_terms() of this light (the LDS copy)
PT_DEV void light_ray(const pt_light& light, vec3 p, vec3& dir, vec3& color, bool cone_terms_staged = false) {
    bool local = light.type == PT_LIGHT_POINT || light.type == PT_LIGHT_SPOT;
    if (local) dir = v3p(light.position) - p;
    else dir = -v3p(light.direction);
    color = v3p(light.color) * light.intensity;
    if (local) {
        float distance = length(dir);
        float falloff = 1.0f;
        if (light.cutoff > 0.0f) falloff = hmax(hmin(1.0f - hpow4(fdiv(distance, light.cutoff)), 1.0f), 0.0f);
        falloff = fdiv(falloff, distance * distance);
        color *= falloff;
    }
    dir = normalize(dir);
    if (light.type == PT_LIGHT_SPOT) {
        float scale, offset;
        if (cone_terms_staged) { float staged[2]; memcpy(staged, light.pad, 8); scale = staged[0]; offset = staged[1]; }
        else spot_cone_terms(light.inner_angle, light.outer_angle, scale, offset);
        float cd = -dot(normalize(v3p(light.direction)), dir);
        float att = saturate(cd * scale + offset);
        att *= att;
        color *= att;
    }
}

// ---------------------------------------------------------------- environment (TextureCube + importance pyramid)
PT_DEV void dir_to_face(vec3 d, int& face, float& u, float& v) {   // D3D major-axis table = inverse of cubemap_to_direction
    float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z), sc, tc, ma;
    if (ax >= ay && ax >= az) { ma = ax; if (d.x >= 0) { face = 0; sc = -d.z; tc = -d.y; } else { face = 1; sc = d.z; tc = -d.y; } }
    else if (ay >= az) { ma = ay; if (d.y >= 0) { face = 2; sc = d.x; tc = d.z; } else { face = 3; sc = d.x; tc = -d.z; } }
    else { ma = az; if (d.z >= 0) { face = 4; sc = d.x; tc = -d.y; } else { face = 5; sc = -d.x; tc = -d.y; } }
    const float rma_ = frcp_refined(ma);
    u = 0.5f * (fdiv_with(sc, ma, rma_) + 1.0f);
    v = 0.5f * (fdiv_with(tc, ma, rma_) + 1.0f);
}
PT_DEV vec3 cube_texel(const uint16_t* cube, int n, int face, int i, int j) {
    const uint2 q = *(const uint2*)(cube + (((size_t)face * n + j) * n + i) * 4);     // 8-B RGBA16F texel
    return {half_bits_to_float((uint16_t)(q.x & 0xffff)), half_bits_to_float((uint16_t)(q.x >> 16)), half_bits_to_float((uint16_t)(q.y & 0xffff))};
}
// Texel a bilinear tap reads: (i, j) on `face`, or -- off the face's edge -- the texel that the tap's direction lands on on the
// neighbouring face (seamless filtering by re-projection, point fetch there).  Address only: the four fetches of a sample are
// issued together afterwards.
PT_DEV size_t cube_tap_index(int n, int face, int i, int j) {
    if (!(i >= 0 && i < n && j >= 0 && j < n)) {
        vec3 d = cubemap_to_direction(face, fdiv((float)i + 0.5f, (float)n), fdiv((float)j + 0.5f, (float)n));
        float u, v;
        dir_to_face(d, face, u, v);
        i = (int)floorf(u * (float)n); j = (int)floorf(v * (float)n);
        i = min(max(i, 0), n - 1); j = min(max(j, 0), n - 1);
    }
    return ((size_t)face * n + j) * n + i;
}
PT_DEV vec3 cube_unpack(uint2 q) {                                       // 8-B RGBA16F texel
    return {half_bits_to_float((uint16_t)(q.x & 0xffff)), half_bits_to_float((uint16_t)(q.x >> 16)), half_bits_to_float((uint16_t)(q.y & 0xffff))};
}
// TextureCube.SampleLevel(linear, dir, 0) (PathTracer.lib.hlsl:700,1042)
PT_DEV vec3 sample_cube(const uint16_t* cube, int n, vec3 d) {
    int face; float u, v;
    dir_to_face(d, face, u, v);
    if (!(u == u) || !(v == v)) return v3(0);
    float x = u * (float)n - 0.5f, y = v * (float)n - 0.5f;
    float fx0 = floorf(x), fy0 = floorf(y);
    float fx = x - fx0, fy = y - fy0;
    int i0 = (int)fx0, j0 = (int)fy0;
    float w00 = (1 - fx) * (1 - fy), w10 = fx * (1 - fy), w01 = (1 - fx) * fy, w11 = fx * fy;
    const size_t a00 = cube_tap_index(n, face, i0, j0), a10 = cube_tap_index(n, face, i0 + 1, j0);
    const size_t a01 = cube_tap_index(n, face, i0, j0 + 1), a11 = cube_tap_index(n, face, i0 + 1, j0 + 1);
    const uint2* texels = (const uint2*)cube;
    const uint2 q00 = gload_u2(texels + a00), q10 = gload_u2(texels + a10), q01 = gload_u2(texels + a01), q11 = gload_u2(texels + a11);
    return cube_unpack(q00) * w00 + cube_unpack(q10) * w10 + cube_unpack(q01) * w01 + cube_unpack(q11) * w11;
}
PT_DEV float imp_load(const EnvRec& e, int level, uint32_t x, uint32_t y) {
    uint32_t n = (uint32_t)e.imp_res >> level;
    if (x >= n || y >= n) return 0.f;
    return e.importance[e.level_offset[level] + (size_t)y * n + x];
}
// One level of SampleImportanceMap (Sampling.hlsli:131-158): pick left/right by the column sums, then upper/lower inside the
// chosen column, re-normalising the random numbers.  sx/sy = 1 for right / lower.
PT_DEV void importance_step(float ul, float ur, float ll, float lr, float& ux, float& uy, uint32_t& sx, uint32_t& sy) {
    const float left = ul + ll, right = ur + lr, total = left + right;
    const float prob_left = fdiv(left, total);
    const bool go_left = ux < prob_left;
    ux = fdiv(go_left ? ux : ux - prob_left, go_left ? prob_left : 1 - prob_left);
    const float prob_upper = fdiv(go_left ? ul : ur, go_left ? left : right);
    const bool up = uy < prob_upper;
    uy = fdiv(up ? uy : uy - prob_upper, up ? prob_upper : 1 - prob_upper);
    sx = go_left ? 0u : 1u;
    sy = up ? 0u : 1u;
}
// The three coarsest level pairs of the blocked pyramid (4^2, 16^2, 64^2: the first 4368 floats, 17 KB) are staged into LDS by the
// shade stage: three of the five dependent fetches of the descent become ds_reads of the same values.
#ifdef PT_LUT_LDS
constexpr int kImpLdsLevels = 3;
constexpr uint32_t kImpLdsFloat4 = (16u + 256u + 4096u) / 4u;
static __shared__ float4 pt_lds_imp[kImpLdsFloat4];
PT_DEV void stage_importance_top(const SceneRec& sc) {   // 256-thread workgroups
    if (sc.has_env) for (uint32_t i = threadIdx.x; i < kImpLdsFloat4; i += 256u) pt_lds_imp[i] = gload_f4((const float4*)sc.env.blocked + i);
    __syncthreads();
}
PT_DEV const float4* importance_lds_top() { return pt_lds_imp; }
// the same staging into LDS a caller owns (kImpLdsFloat4 float4; the traversal stages lend their stack memory to the pre-pass)
PT_DEV void stage_importance_top_into(const SceneRec& sc, float4* dst) {
    if (sc.has_env) for (uint32_t i = threadIdx.x; i < kImpLdsFloat4; i += 256u) dst[i] = gload_f4((const float4*)sc.env.blocked + i);
    __syncthreads();
}
#else
constexpr int kImpLdsLevels = 0;
PT_DEV void stage_importance_top(const SceneRec&) {}
PT_DEV const float4* importance_lds_top() { return nullptr; }
#endif
// `lds_top`: the three coarsest level pairs in LDS (importance_lds_top() or a caller's staging), unused when kImpLdsLevels == 0
PT_DEV vec2 sample_importance_map(const EnvRec& e, float ux, float uy, float& pdf, const float4* lds_top) {    // Sampling.hlsli:123-163
    // The reference descends ten levels with four dependent point loads each.  Here one 64-B fetch of a 4x4 block of the
    // finer level of a pair serves two levels: the coarser level's 2x2 values are re-summed from the block in the order the
    // pyramid build uses (k_importance_level: ((ul + ll) + ur) + lr), which reproduces the stored sums bit for bit.
#ifdef PT_PROBE_NO_DESCENT      // PROBE ONLY: what the five-level descent costs the shade stage (wrong sampling, timing only)
    pdf = 1.0f;
    return {ux, uy};
#endif
    uint32_t px = 0, py = 0;
    float value = 0.f;
#pragma unroll 1
    for (int k = 0; k < 5; k++) {
        const uint32_t nb = 1u << (2 * k);                                 // blocks per row of this pair's finer level
        const size_t block = (size_t)py * nb + px;
        float4 r0, r1, r2, r3;
        if (k < kImpLdsLevels) { const float4* blk = lds_top + e.blocked_offset[k] / 4u + block * 4u; r0 = blk[0]; r1 = blk[1]; r2 = blk[2]; r3 = blk[3]; }
        else { const float4* blk = (const float4*)(e.blocked + e.blocked_offset[k]) + block * 4; r0 = blk[0]; r1 = blk[1]; r2 = blk[2]; r3 = blk[3]; }
        const float a_ul = ((r0.x + r1.x) + r0.y) + r1.y, a_ur = ((r0.z + r1.z) + r0.w) + r1.w;
        const float a_ll = ((r2.x + r3.x) + r2.y) + r3.y, a_lr = ((r2.z + r3.z) + r2.w) + r3.w;
        uint32_t sx, sy, tx, ty;
        importance_step(a_ul, a_ur, a_ll, a_lr, ux, uy, sx, sy);
        const float4 top = sy ? r2 : r0, bot = sy ? r3 : r1;              // the chosen texel's 2x2 children
        const float ul = sx ? top.z : top.x, ur = sx ? top.w : top.y, ll = sx ? bot.z : bot.x, lr = sx ? bot.w : bot.y;
        importance_step(ul, ur, ll, lr, ux, uy, tx, ty);
        value = ty ? (tx ? lr : ll) : (tx ? ur : ul);
        px = (px << 2) | (sx << 1) | tx;
        py = (py << 2) | (sy << 1) | ty;
    }
    float w = (float)e.imp_res;
    pdf = fdiv(w * w * value, e.imp_total);                          // value = level-0 texel (px, py); imp_total = mips[10][0]
    return {fdiv((float)px + ux, w), fdiv((float)py + uy, w)};      // both axes / width (quirk q10)
}
// SampleEnvironmentMap's hit-independent half (PathTracer.lib.hlsl:688-703): direction, solid-angle pdf and radiance of the sample
// the random numbers (u0, u1) pick.  It depends on the pixel's random sequence only, never on the hit, which is what lets the
// wavefront pipeline draw it ahead of the shade stage (pt_wavefront.hip env_prepass).
struct EnvSample { vec3 dir; float pdf; vec3 color; };
PT_DEV EnvSample environment_light_sample(const SceneRec& sc, float environment_intensity, float u0, float u1, const float4* lds_top) {
    EnvSample e;
    e.pdf = 1; e.dir = v3(0, 0, 1); e.color = v3(0);
    if (sc.has_env) {
        vec2 uv = sample_importance_map(sc.env, u0, u1, e.pdf, lds_top);
        e.dir = square_to_sphere(uv_to_square(uv));
        e.pdf = fdiv(e.pdf, 4 * kPi);
        e.color = environment_intensity * sample_cube(sc.env.cube, sc.env.cube_n, e.dir);
    }
    return e;
}
PT_DEV float importance_map_pdf(const EnvRec& e, vec2 uv) {                                                   // Sampling.hlsli:165-174, Common.hlsli:12-15
    float total = e.imp_total;
    float r = (float)e.imp_res;
    int px = f2i(floorf(uv.x * r) - .5f), py = f2i(floorf(uv.y * r) - .5f);       // UVToPixel: off by one (quirk q9)
    float value = (px < 0 || py < 0) ? 0.f : imp_load(e, 0, (uint32_t)px, (uint32_t)py);
    return fdiv(r * r * value, total);
}

// ---------------------------------------------------------------- misc
PT_DEV vec3 offset_ray(vec3 p, vec3 ng) {                           // PathTracer.lib.hlsl:260-268 (RT Gems ch. 6)
    const float origin = 1.0f / 32.0f, float_scale = 1.0f / 65536.0f, int_scale = 256.0f;
    int ox = f2i(int_scale * ng.x), oy = f2i(int_scale * ng.y), oz = f2i(int_scale * ng.z);
    float ix = __int_as_float(__float_as_int(p.x) + (p.x < 0 ? -ox : ox));
    float iy = __int_as_float(__float_as_int(p.y) + (p.y < 0 ? -oy : oy));
    float iz = __int_as_float(__float_as_int(p.z) + (p.z < 0 ? -oz : oz));
    return {fabsf(p.x) < origin ? p.x + float_scale * ng.x : ix, fabsf(p.y) < origin ? p.y + float_scale * ng.y : iy,
            fabsf(p.z) < origin ? p.z + float_scale * ng.z : iz};
}
PT_DEV float luminance(vec3 c) { return dot(c, v3(0.2126f, 0.7152f, 0.0722f)); }   // Color.hlsli:4-7

}  // namespace pt
