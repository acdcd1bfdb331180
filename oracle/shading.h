// oracle/shading.h -- TEST INFRASTRUCTURE ONLY (CPU oracle; see oracle/README.md).
//
// Literal CPU restatement of the reference's shading headers, function by function:
//   Source/Shaders/Random.hlsli, Common.hlsli, Vertex.hlsli, Transforms.hlsli, Color.hlsli,
//   Lights.hlsli, Bsdf.hlsli, Sampling.hlsli and the lobe logic of PathTracer.lib.hlsl:383-667.
// Every function cites the file:line it follows.  PARITY UNPINNED: the reference ships no golden
// vectors or tests for this path; the only reference-supplied fixture is
// the Sheen_E LUT.  These functions are pinned by analytic known-answer tests in tests/.
#pragma once
#include "hlsl.h"

namespace orc {
using namespace hlsl;

// ---------------------------------------------------------------- Random.hlsli:17-30
inline uint4 pcg4d(uint4 v) {
    v.x = v.x * 1664525u + 1013904223u; v.y = v.y * 1664525u + 1013904223u;
    v.z = v.z * 1664525u + 1013904223u; v.w = v.w * 1664525u + 1013904223u;
    v.x += v.y * v.w; v.y += v.z * v.x; v.z += v.x * v.y; v.w += v.y * v.z;
    v.x ^= v.x >> 16u; v.y ^= v.y >> 16u; v.z ^= v.z >> 16u; v.w ^= v.w >> 16u;
    v.x += v.y * v.w; v.y += v.z * v.x; v.z += v.x * v.y; v.w += v.y * v.z;
    return v;
}
// Random.hlsli:3-15
inline void pcg3d(uint32_t v[3]) {
    for (int i = 0; i < 3; i++) v[i] = v[i] * 1664525u + 1013904223u;
    v[0] += v[1] * v[2]; v[1] += v[2] * v[0]; v[2] += v[0] * v[1];
    for (int i = 0; i < 3; i++) v[i] ^= v[i] >> 16u;
    v[0] += v[1] * v[2]; v[1] += v[2] * v[0]; v[2] += v[0] * v[1];
}
// PathTracer.lib.hlsl:144-148.  The literal 4294967295.0 is 2^32 in fp32 (quirk q29): uint->float
// rounds to nearest even, then an fp32 divide by 2^32; u == 1.0f is reachable.
inline float4 GenerateNextRandom(uint32_t px, uint32_t py, uint32_t seed, int& count) {
    uint4 r = pcg4d(uint4{px, py, seed, (uint32_t)count});
    count++;
    const float d = 4294967296.0f;
    return {(float)r.x / d, (float)r.y / d, (float)r.z / d, (float)r.w / d};
}

// ---------------------------------------------------------------- Common.hlsli
inline int2 UVToPixel(float2 uv, int2 res) {                      // :12-15 (off by one, quirk q9)
    return {f2i(floorf(uv.x * (float)res.x) - .5f), f2i(floorf(uv.y * (float)res.y) - .5f)};
}
inline float2 PixelToUV(int2 p, int2 res) {                        // :18-21
    return {((float)p.x + .5f) / (float)res.x, ((float)p.y + .5f) / (float)res.y};
}
inline void CreateBasis(float3 n, float3& t, float3& b) {          // :33-42
    if (fabsf(n.x) > fabsf(n.z)) b = {-n.y, n.x, 0};
    else b = {0, -n.z, n.y};
    b = normalize(b);
    t = cross(b, n);
}
inline void CreateBasisAccurate(float3 n, float3& b1, float3& b2) {  // :46-53
    float sign = n.z >= 0.0f ? 1.0f : -1.0f;
    const float a = -1.0f / (sign + n.z);
    const float b = n.x * n.y * a;
    b1 = {1.0f + sign * n.x * n.x * a, sign * b, -sign * n.x};
    b2 = {b, sign + n.y * n.y * a, -n.y};
}
inline float2 SignNotZero(float2 v) { return {v.x >= 0 ? 1.f : -1.f, v.y >= 0 ? 1.f : -1.f}; }  // :68-74
inline float2 EncodeOctahedralMap(float3 n) {                     // :76-88
    float3 o = n / (fabsf(n.x) + fabsf(n.y) + fabsf(n.z));
    if (o.z >= 0.f) return {o.x, o.y};
    float2 s = SignNotZero({o.x, o.y});
    return {s.x * (1.f - fabsf(o.y)), s.y * (1.f - fabsf(o.x))};
}
inline float3 DecodeOctahedralMap(float2 e) {                     // :90-103
    float3 r;
    r.z = 1.f - fabsf(e.x) - fabsf(e.y);
    if (r.z >= 0.f) { r.x = e.x; r.y = e.y; }
    else { float2 s = SignNotZero(e); r.x = s.x * (1.f - fabsf(e.y)); r.y = s.y * (1.f - fabsf(e.x)); }
    return normalize(r);
}

// ---------------------------------------------------------------- Vertex.hlsli
inline float4 UnpackR10G10B10A2(uint32_t p) {                      // :46-50
    return {(float)(p & 0x3ff) / 1023.f, (float)((p >> 10) & 0x3ff) / 1023.f,
            (float)((p >> 20) & 0x3ff) / 1023.f, (float)((p >> 30) & 0x3) / 3.f};
}
inline void DecodeTangentSpace(float4 enc, float3& normal, float4& tangent) {   // :5-19
    normal = DecodeOctahedralMap(float2{enc.x, enc.y} * 2 - 1);
    float3 ct, cb;
    CreateBasisAccurate(normal, ct, cb);
    float angle = TAU * enc.z;                                    // no -0.5: tangent comes out negated (quirk q25)
    float3 t = o_cos(angle) * ct + o_sin(angle) * cb;
    tangent = {t.x, t.y, t.z, enc.w > 0 ? 1.f : -1.f};
}
// GPU-side encoder, Vertex.hlsli:21-44 (used by Skin.cs.hlsl:134).  No clamp on the tangent.
inline uint32_t EncodeTangentSpaceShader(float3 normal, float4 tangent) {
    float2 en = 0.5f * EncodeOctahedralMap(normal) + 0.5f;
    uint32_t qx = f2u(clamp(en.x, 0, 1) * 1023 + 0.5f), qy = f2u(clamp(en.y, 0, 1) * 1023 + 0.5f);
    float2 un = {(float)qx / 1023.0f, (float)qy / 1023.0f};
    normal = DecodeOctahedralMap(2.0f * un - 1.0f);
    float3 ct, cb;
    CreateBasisAccurate(normal, ct, cb);
    float3 t3 = {tangent.x, tangent.y, tangent.z};
    float angle = o_atan2(dot(t3, cb), dot(t3, ct));
    float enc_t = (angle / TAU) + 0.5f;
    uint32_t qt = f2u(enc_t * 1023 + 0.5f);
    uint32_t qw = tangent.w == 1 ? 3u : 0u;
    return qx | (qy << 10) | (qt << 20) | (qw << 30);
}
// CPU-side encoders, Source/Gltf.cpp:65-104 (tangent clamped; EncodeNormal leaves angle bits 0).
inline uint32_t EncodeTangentSpaceHost(float3 normal, float4 tangent) {
    float2 en = 0.5f * EncodeOctahedralMap(normal) + 0.5f;
    uint32_t qx = f2u(clamp(en.x, 0, 1) * 1023.0f + 0.5f), qy = f2u(clamp(en.y, 0, 1) * 1023.0f + 0.5f);
    float2 un = {(float)qx / 1023.0f, (float)qy / 1023.0f};
    normal = DecodeOctahedralMap(2.0f * un - 1.0f);
    float3 ct, cb;
    CreateBasisAccurate(normal, ct, cb);                           // Gltf.cpp:57-63 is the same basis
    float3 t3 = {tangent.x, tangent.y, tangent.z};
    float angle = o_atan2(dot(t3, cb), dot(t3, ct));
    float enc_t = (angle / 6.283185307179586f) + 0.5f;
    uint32_t qt = f2u(clamp(enc_t, 0, 1) * 1023.0f + 0.5f);
    uint32_t qw = tangent.w == 1.0f ? 3u : 0u;
    return qx | (qy << 10) | (qt << 20) | (qw << 30);
}
inline uint32_t EncodeNormalHost(float3 normal) {
    float2 en = 0.5f * EncodeOctahedralMap(normal) + 0.5f;
    uint32_t qx = f2u(clamp(en.x, 0, 1) * 1023.0f + 0.5f), qy = f2u(clamp(en.y, 0, 1) * 1023.0f + 0.5f);
    return qx | (qy << 10) | (3u << 30);
}

// ---------------------------------------------------------------- Transforms.hlsli
inline float3 CubemapToDirection(int face, float2 uv) {            // :10-50
    float3 ud, vd, fd;
    switch (face) {
        case 0: fd = {1, 0, 0}; ud = {0, 0, -1}; vd = {0, -1, 0}; break;
        case 1: fd = {-1, 0, 0}; ud = {0, 0, 1}; vd = {0, -1, 0}; break;
        case 2: fd = {0, 1, 0}; ud = {1, 0, 0}; vd = {0, 0, 1}; break;
        case 3: fd = {0, -1, 0}; ud = {1, 0, 0}; vd = {0, 0, -1}; break;
        case 4: fd = {0, 0, 1}; ud = {1, 0, 0}; vd = {0, -1, 0}; break;
        default: fd = {0, 0, -1}; ud = {-1, 0, 0}; vd = {0, -1, 0}; break;
    }
    uv = uv * 2 - 1;
    return normalize(fd + uv.x * ud + uv.y * vd);
}
inline float2 UvToUnitSquare(float2 uv) { return uv * float2{2, -2} + float2{-1, 1}; }            // :52-55
inline float2 UnitSquareToUv(float2 s) { return (s - float2{-1, 1}) * float2{0.5f, -0.5f}; }      // :57-60
inline float2 SquareToDisk2(float2 s) {                             // :83-90
    float r = hmax(fabsf(s.x), fabsf(s.y));
    float phi = r == 0 ? 0 : (PI * (r + (fabsf(s.y) - fabsf(s.x))) / (4 * r));
    return {sign(s.x) * r * o_cos(phi), sign(s.y) * r * o_sin(phi)};
}
inline float3 SquareToSphere(float2 s) {                            // :124-136
    float d = 1 - (fabsf(s.x) + fabsf(s.y));
    float r = 1 - fabsf(d);
    float phi = (r == 0) ? 0 : (PI / 4) * ((fabsf(s.y) - fabsf(s.x)) / r + 1);
    float f = r * sqrtf(2 - r * r);
    return {f * sign(s.x) * o_cos(phi), f * sign(s.y) * o_sin(phi), sign(d) * (1 - r * r)};
}
inline float2 SphereToSquare(float3 p) {                            // :138-149
    float r = sqrtf(1 - fabsf(p.z));
    float phi = o_atan2(fabsf(p.y), fabsf(p.x));
    float d = sign(p.z) * (1 - r);
    float diff = r * ((4 / PI) * phi - 1);
    return {sign(p.x) * 0.5f * (1 - d - diff), sign(p.y) * 0.5f * (1 - d + diff)};
}

// ---------------------------------------------------------------- Color.hlsli
inline float Luminance(float3 c) { return dot(c, float3{0.2126f, 0.7152f, 0.0722f}); }          // :4-7
inline float3 EncodeSrgb(float3 c) {                                // :9-17
    auto f = [](float x) { return x <= 0.0031308f ? x * 12.92f : 1.055f * hpow(x, 1.f / 2.4f) - 0.055f; };
    return {f(c.x), f(c.y), f(c.z)};
}

// ---------------------------------------------------------------- Lights.hlsli
struct Light {                                                      // :9-19 (64 B)
    int type; float3 position; float cutoff; float3 direction; float intensity; float3 color;
    float inner_angle; float outer_angle; float pad[2];
};
static_assert(sizeof(Light) == 64, "Light");
struct LightRay { float3 direction; float3 color; };
inline LightRay GetLightRay(const Light& light, float3 p) {         // :26-61
    LightRay ray;
    if (light.type == 0 || light.type == 1) ray.direction = light.position - p;
    else ray.direction = -light.direction;
    ray.color = light.color * light.intensity;
    if (light.type == 0 || light.type == 1) {
        float distance = length(ray.direction);
        float falloff = 1.0f;
        if (light.cutoff > 0.0f) falloff = hmax(hmin(1.0f - hpow4(distance / light.cutoff), 1.0f), 0.0f);
        falloff /= distance * distance;
        ray.color *= falloff;
    }
    ray.direction = normalize(ray.direction);
    if (light.type == 1) {
        float scale = 1.0f / hmax(0.001f, o_cos(light.inner_angle) - o_cos(light.outer_angle));
        float offset = -o_cos(light.outer_angle) * scale;
        float cd = -dot(normalize(light.direction), ray.direction);
        float att = saturate(cd * scale + offset);
        att *= att;
        ray.color *= att;
    }
    return ray;
}

// ---------------------------------------------------------------- Bsdf.hlsli
struct SurfaceProperties {                                          // :4-24 (36 floats)
    float3 albedo; float alpha; float metalness; float2 roughness_squared; float3 shading_normal;
    float3 anisotropy_tangent; float3 anisotropy_bitangent; float ior; float3 specular_color;
    float specular_factor; float clearcoat; float clearcoat_roughness; float3 clearcoat_normal;
    float3 sheen_color; float sheen_roughness_squared; float transmissive; float thickness;
    float attenuation_distance; float3 attenuation_color;
};
static_assert(sizeof(SurfaceProperties) == 36 * 4, "SurfaceProperties");
static const float MINIMUM_ROUGHNESS = 0.001f;                      // :26

inline float Heavyside(float a) { return a > 0 ? 1.f : 0.f; }       // :29-32
inline float MaxValue(float3 c) { return hmax(hmax(c.x, c.y), c.z); }  // :34-37
inline float SchlickFresnel(float f0, float ndv) { return f0 + (1 - f0) * hpow5(1 - fabsf(ndv)); }      // :39-42
inline float3 SchlickFresnel(float3 f0, float ndv) { return f0 + (1 - f0) * hpow5(1 - fabsf(ndv)); }   // :44-47
inline float GgxD(float a, float ndh) {                             // :50-57
    float a2 = a * a;
    float num = a2 * Heavyside(ndh);
    float den = ndh * ndh * (a2 - 1) + 1;
    den *= PI * den;
    return num / den;
}
inline float GgxCorrelatedV(float a, float ndl, float ndv, float hdl, float hdv) {   // :78-85
    float a2 = a * a;
    float num = 0.5f * Heavyside(hdl) * Heavyside(hdv);
    float den = fabsf(ndv) * sqrtf(a2 + (1 - a2) * ndl * ndl);
    den += fabsf(ndl) * sqrtf(a2 + (1 - a2) * ndv * ndv);
    return num / den;
}
inline float SpecularBrdf(float a, float ndl, float ndv, float ndh, float hdl, float hdv) {  // :87-90
    return GgxCorrelatedV(a, ndl, ndv, hdl, hdv) * GgxD(a, ndh);
}
inline float GgxAnisotropicD(float2 a, float3 h) {                  // :93-99
    float a2 = a.x * a.y;
    float3 f = {a.y * h.x, a.x * h.y, a2 * h.z};
    float w2 = a2 / dot(f, f);
    return Heavyside(h.z) * a2 * w2 * w2 / PI;
}
inline float GgxAnisotropicCorrelatedV(float2 a, float3 v, float3 l, float hdv, float hdl) {  // :117-123
    float num = 0.5f * Heavyside(hdv) * Heavyside(hdl);
    float vv = fabsf(l.z) * length(float3{a.x * v.x, a.y * v.y, v.z});
    float ll = fabsf(v.z) * length(float3{a.x * l.x, a.y * l.y, l.z});
    return num / (vv + ll);
}
inline float AnisotropicSpecularBrdf(float2 a, float3 v, float3 h, float3 l) {   // :125-130
    float hdv = dot(h, v), hdl = dot(h, l);
    return GgxAnisotropicCorrelatedV(a, v, l, hdv, hdl) * GgxAnisotropicD(a, h);
}
inline float3 LambertDiffuse(float3 c) { return c / PI; }           // :132-135
inline float3 FresnelMix(float3 f0_color, float ior, float weight, float3 base, float3 layer, float hdv) {  // :137-144
    float3 f0 = F3((1 - ior) / (1 + ior));
    f0 *= f0 * f0_color;
    f0 = hmin(f0, F3(1));
    float3 fr = SchlickFresnel(f0, hdv);
    return (1 - weight * MaxValue(fr)) * base + weight * fr * layer;
}
inline float3 ConductorFresnel(float3 specular, float3 f0, float hdv) { return specular * SchlickFresnel(f0, hdv); }  // :146-149
inline float3 FresnelCoat(float ior, float weight, float3 base, float3 layer, float ndv) {   // :157-163
    float f0 = (1 - ior) / (1 + ior);
    f0 *= f0;
    float fr = SchlickFresnel(f0, ndv);
    return lerp(base, layer, weight * fr);
}
inline float SheenNormalDistribution(float alpha, float ndh) {      // :166-173
    float inv_r = 1 / alpha;
    float cos2h = ndh * ndh;
    float sin2h = 1 - cos2h;
    return (2 + inv_r) * hpow(sin2h, inv_r * 0.5f) / (2 * PI);
}
inline float SheenL(float alpha, float x) {                         // :175-184
    float t = (1 - alpha) * (1 - alpha);
    float a = lerp(21.5473f, 25.3245f, t);
    float b = lerp(3.82987f, 3.32435f, t);
    float c = lerp(0.19823f, 0.16801f, t);
    float d = lerp(-1.97760f, -1.27393f, t);
    float e = lerp(-4.32054f, -4.85967f, t);
    return a / (1 + b * hpow(x, c)) + d * x + e;
}
inline float SheenShadowing(float alpha, float c) {                 // :186-193
    if (c < 0.5f) return o_exp(SheenL(alpha, c));
    return o_exp(2 * SheenL(alpha, 0.5f) - SheenL(alpha, 1 - c));
}
inline float SheenVisibility(float alpha, float ndl, float ndv) {   // :195-198
    return clamp(1 / ((1 + SheenShadowing(alpha, ndl) + SheenShadowing(alpha, ndv)) * 4 * ndl * ndv), 0, 1);
}
inline float SheenBrdf(float alpha, float ndl, float ndv, float ndh) {  // :200-203 (note the swapped args)
    return SheenNormalDistribution(alpha, ndh) * SheenVisibility(alpha, ndv, ndl);
}
// Bsdf.hlsli:204-208: Sheen_E LUT, SampleLevel(linear_clamp, (cos_theta, alpha), 0) on a 16x16 R16F.
struct SheenLut { float v[16 * 16]; };
inline float SheenE(const SheenLut& lut, float alpha, float cos_theta) {
    auto tap = [&](int i, int j) {
        i = i < 0 ? 0 : (i > 15 ? 15 : i); j = j < 0 ? 0 : (j > 15 ? 15 : j);
        return lut.v[j * 16 + i];
    };
    float x = cos_theta * 16.f - 0.5f, y = alpha * 16.f - 0.5f;
    if (!(x == x)) x = 0; if (!(y == y)) y = 0;
    x = clamp(x, -1.f, 16.f); y = clamp(y, -1.f, 16.f);
    float fx0 = floorf(x), fy0 = floorf(y);
    int i0 = (int)fx0, j0 = (int)fy0;
    float fx = x - fx0, fy = y - fy0;
    float w00 = (1 - fx) * (1 - fy), w10 = fx * (1 - fy), w01 = (1 - fx) * fy, w11 = fx * fy;
    return tap(i0, j0) * w00 + tap(i0 + 1, j0) * w10 + tap(i0, j0 + 1) * w01 + tap(i0 + 1, j0 + 1) * w11;
}
inline float3 SheenMix(const SheenLut& lut, float3 material, float3 layer, float3 sheen_color, float alpha, float ndl, float ndv) {  // :210-214
    float scaling = hmin(1.0f - MaxValue(sheen_color) * SheenE(lut, alpha, ndv), 1.0f - MaxValue(sheen_color) * SheenE(lut, alpha, ndl));
    return sheen_color * layer + material * scaling;
}
inline float ModulateRoughness(float a, float ior) {                // :216-220
    return clamp(lerp(0, a, saturate(2 * (ior - 1))), MINIMUM_ROUGHNESS, 1.0f);
}
inline float3 ThinSurfaceTransmissionBtdf(float3 color, float a, float ior, float3 n, float3 v, float3 l) {  // :222-228
    a = ModulateRoughness(a, ior);
    l = l - 2 * dot(n, l) * n;
    float3 h = normalize(v + l);
    return color * SpecularBrdf(a, dot(n, l), dot(n, v), dot(n, h), dot(h, l), dot(h, v));
}
// Bsdf.hlsli:241-282 (has_flag=false) and :284-325 (has_flag=true, is_transmission given).
inline float3 GltfBsdf(const SheenLut& lut, SurfaceProperties sp, float3 v, float3 l, bool has_flag, bool is_transmission) {
    float2 a = sp.roughness_squared;
    float3 n = sp.shading_normal;
    float3 h = normalize(v + l);
    float3x3 w2t = M3(sp.anisotropy_tangent, sp.anisotropy_bitangent, sp.shading_normal);
    float3 vl = mul(w2t, v), hl = mul(w2t, h), ll = mul(w2t, l);
    float hdl = dot(h, l), hdv = dot(h, v);
    float h_dot_abs_l = dot(normalize(float3{ll.x, ll.y, fabsf(ll.z)} + vl), vl);
    bool refl = !has_flag || !is_transmission;       // terms kept when reflecting
    bool trans = !has_flag || is_transmission;       // terms kept when transmitting
    float3 specular = refl ? saturate(ll.z) * F3(AnisotropicSpecularBrdf(a, vl, hl, ll)) : F3(0);
    float3 diffuse = refl ? saturate(ll.z) * LambertDiffuse(sp.albedo) : F3(0);
    float3 transmission = trans ? saturate(-ll.z) * ThinSurfaceTransmissionBtdf(sp.albedo, a.y, sp.ior, n, v, l) : F3(0);
    diffuse = lerp(diffuse, transmission, sp.transmissive);
    float3 dialectric = FresnelMix(sp.specular_color, sp.ior, sp.specular_factor, diffuse, specular, h_dot_abs_l);
    float3 metal = refl ? ConductorFresnel(specular, sp.albedo, hdv) : F3(0);
    float3 material = lerp(dialectric, metal, sp.metalness);
    sp.sheen_roughness_squared = clamp(sp.sheen_roughness_squared, 0.000001f, 1);
    float3 sheen_brdf = refl ? F3(saturate(ll.z) * SheenBrdf(sp.sheen_roughness_squared, ll.z, vl.z, hl.z)) : F3(0);
    material = SheenMix(lut, material, sheen_brdf, sp.sheen_color, sp.sheen_roughness_squared, ll.z, vl.z);
    float cndv = dot(n, v), cndh = dot(n, h), cndl = dot(n, l);     // (sic) shading normal, not clearcoat normal
    float cc = refl ? saturate(cndl) * SpecularBrdf(sp.clearcoat_roughness, cndl, cndv, cndh, hdl, hdv) : 0;
    return FresnelCoat(1.5f, sp.clearcoat, material, F3(cc), cndv);
}

// ---------------------------------------------------------------- Sampling.hlsli
inline float3 SampleCosineWeightedHemisphereLocal(float2 u) {      // :16-22
    float2 d = SquareToDisk2(UvToUnitSquare(u));
    return {d.x, d.y, sqrtf(1 - d.x * d.x - d.y * d.y)};
}
inline float3 SampleCosineWeightedHemisphere(float3 n, float2 u) {  // :26-33
    float theta = TAU * u.x;
    u.y = 2 * u.y - 1;
    float s = sqrtf(1.0f - u.y * u.y);
    float3 sphere = {s * o_cos(theta), s * o_sin(theta), u.y};
    return normalize(n + sphere);
}
inline float CosineWeightedHemispherePdf(float3 n, float3 v) { return saturate(dot(v, n) / PI); }   // :35-38
inline float3 SampleGgxNormal(float a, float2 u) {                  // :41-52
    float phi = TAU * u.x;
    float cos_theta = sqrtf((1 - u.y) / (1 + (a * a - 1) * u.y));
    float sin_theta = sqrtf(1 - cos_theta * cos_theta);
    return {sin_theta * o_cos(phi), sin_theta * o_sin(phi), cos_theta};
}
inline float GgxNormalPdf(float a, float3 n, float3 h) { float ndh = dot(n, h); return GgxD(a, ndh) * ndh; }  // :54-58
inline float3 SampleGgxAnisotropicNormal(float2 a, float2 u) {      // :60-65
    float3 h = SampleCosineWeightedHemisphereLocal(u);
    h.x *= a.x; h.y *= a.y;
    return normalize(h);
}
inline float GgxAnisotropicNormalPdf(float2 a, float3 hl) { return GgxAnisotropicD(a, hl) * hl.z; }  // :67-70

// ---------------------------------------------------------------- PathTracer.lib.hlsl:383-667
inline float BalanceHeuristic(float pdf, float other) { return pdf / (pdf + other); }   // :383-386

inline float3 SampleClearcoat(const SurfaceProperties& sp, float3 v, float2 u) {   // :394-406
    float3 n = sp.clearcoat_normal, t, b;
    CreateBasis(n, t, b);
    float3x3 l2w = transpose(M3(t, b, n));
    float3 h = mul(l2w, SampleGgxNormal(sp.clearcoat_roughness, u));
    return reflect(-v, h);
}
inline float ClearcoatPdf(const SurfaceProperties& sp, float3 v, float3 l) {       // :408-416
    float3 h = normalize(v + l);
    float pdf = GgxNormalPdf(sp.clearcoat_roughness, sp.clearcoat_normal, h);
    pdf /= 4 * dot(v, h);
    return pdf;
}
inline float3 SampleSpecular(const SurfaceProperties& sp, float3 v, float2 u) {    // :428-442
    float3x3 l2w = transpose(M3(sp.anisotropy_tangent, sp.anisotropy_bitangent, sp.shading_normal));
    float3 h = mul(l2w, SampleGgxAnisotropicNormal(sp.roughness_squared, u));
    return reflect(-v, h);
}
inline float SpecularPdf(const SurfaceProperties& sp, float3 v, float3 l) {        // :444-460
    float3 h = normalize(v + l);
    float3x3 w2l = M3(sp.anisotropy_tangent, sp.anisotropy_bitangent, sp.shading_normal);
    float3 lh = mul(w2l, h);
    float pdf = GgxAnisotropicNormalPdf(sp.roughness_squared, lh);
    pdf /= 4 * dot(v, h);
    return pdf;
}
inline float3 SampleTransmission(const SurfaceProperties& sp, float3 v, float2 u) {  // :472-487
    float3 n = sp.shading_normal;
    float3x3 l2w = transpose(M3(sp.anisotropy_tangent, sp.anisotropy_bitangent, n));
    float a = ModulateRoughness(sp.roughness_squared.y, sp.ior);
    float3 h = mul(l2w, SampleGgxNormal(a, u));
    float3 l = reflect(-v, h);
    l = l - 2 * dot(n, l) * n;
    return l;
}
inline float TransmissionPdf(const SurfaceProperties& sp, float3 v, float3 l) {    // :489-500
    float a = ModulateRoughness(sp.roughness_squared.y, sp.ior);
    float3 n = sp.shading_normal;
    l = l - 2 * dot(n, l) * n;
    float3 h = normalize(v + l);
    float pdf = GgxNormalPdf(a, n, h);
    pdf /= 4 * dot(v, h);
    return pdf;
}
enum BsdfLayer { LAYER_DIFFUSE, LAYER_SPECULAR, LAYER_SHEEN, LAYER_CLEARCOAT, LAYER_ALPHA, LAYER_TRANSMISSION };  // :502-509
struct LayerProbs { float alpha, clearcoat, sheen, specular, diffuse, transmission; };
inline BsdfLayer SelectBsdf(float u, const LayerProbs& p) {         // :511-533
    if (u <= p.alpha) return LAYER_ALPHA;
    u -= p.alpha;
    if (u <= p.clearcoat) return LAYER_CLEARCOAT;
    u -= p.clearcoat;
    if (u <= p.sheen) return LAYER_SHEEN;
    u -= p.sheen;
    if (u <= p.specular) return LAYER_SPECULAR;
    u -= p.specular;
    if (u <= p.transmission) return LAYER_TRANSMISSION;
    return LAYER_DIFFUSE;
}
inline LayerProbs LayerProbabilities(const SurfaceProperties& sp, float3 v) {      // :535-553
    LayerProbs p;
    float remaining = 1;
    p.alpha = 1.0f - sp.alpha;
    remaining -= p.alpha;
    p.clearcoat = FresnelCoat(1.5f, sp.clearcoat, F3(0), F3(1), dot(sp.clearcoat_normal, v)).x;
    p.clearcoat *= remaining;
    remaining -= p.clearcoat;
    p.sheen = any_gt0(sp.sheen_color) ? 0.5f : 0.0f;
    p.sheen *= remaining;
    remaining -= p.sheen;
    p.specular = 0.5f;
    p.specular *= remaining;
    remaining -= p.specular;
    p.transmission = sp.transmissive;
    p.transmission *= remaining;
    remaining -= p.transmission;
    p.diffuse = remaining;
    return p;
}
inline float BsdfPdf(const SurfaceProperties& sp, float3 v, float3 l, bool is_transmission, const LayerProbs& p) {  // :555-565
    if (is_transmission) return p.transmission * TransmissionPdf(sp, v, l);
    float pdf = p.clearcoat * ClearcoatPdf(sp, v, l);
    pdf += p.sheen * CosineWeightedHemispherePdf(sp.shading_normal, l);     // SheenPdf :423-426
    pdf += p.specular * SpecularPdf(sp, v, l);
    pdf += p.diffuse * CosineWeightedHemispherePdf(sp.shading_normal, l);   // DiffusePdf :467-470
    return pdf;
}
struct ShadingEnv { const SheenLut* lut; uint32_t flags; };
enum {
    F_NONE = 1 << 0, F_CULL_BACKFACE = 1 << 1, F_ACCUMULATE = 1 << 2, F_LUMINANCE_CLAMP = 1 << 3,
    F_INDIRECT_ENVIRONMENT_ONLY = 1 << 4, F_POINT_LIGHTS = 1 << 5, F_SHADOW_RAYS = 1 << 6, F_ALPHA_SHADOWS = 1 << 7,
    F_ENVIRONMENT_MAP = 1 << 8, F_ENVIRONMENT_MIS = 1 << 9, F_MATERIAL_DIFFUSE_WHITE = 1 << 10,
    F_MATERIAL_USE_GEOMETRIC_NORMALS = 1 << 11, F_MATERIAL_MIS = 1 << 12, F_SHOW_NAN = 1 << 13, F_SHOW_INF = 1 << 14,
    F_SHADING_NORMAL_ADAPTATION = 1 << 15
};                                                                  // PathTracer.lib.hlsl:74-91
inline float3 EvaluateBsdf(const ShadingEnv& env, const SurfaceProperties& sp, float3 ng, float3 v, float3 l, float& pdf) {  // :567-593
    if (env.flags & F_MATERIAL_DIFFUSE_WHITE) {
        float ndl = saturate(dot(sp.shading_normal, l));
        pdf = ndl / PI;
        return F3(ndl / PI);
    }
    if (env.flags & F_MATERIAL_MIS) {
        bool is_transmission = (dot(ng, l) * dot(ng, v)) < 0;
        LayerProbs p = LayerProbabilities(sp, v);
        pdf = BsdfPdf(sp, v, l, is_transmission, p);
        return sp.alpha * GltfBsdf(*env.lut, sp, v, l, true, is_transmission);
    }
    float ndl = saturate(dot(sp.shading_normal, l));
    pdf = ndl / PI;
    pdf *= sp.alpha;
    return sp.alpha * GltfBsdf(*env.lut, sp, v, l, false, false);
}
inline float3 SampleBsdf(const ShadingEnv& env, const SurfaceProperties& sp, float3 u, float3 v, float3& l, float& pdf,
                         bool& is_transmission, bool& use_mis) {    // :595-667
    if (env.flags & F_MATERIAL_DIFFUSE_WHITE) {
        use_mis = true;
        is_transmission = false;
        float3 n = sp.shading_normal;
        l = SampleCosineWeightedHemisphere(n, {u.y, u.z});
        pdf = CosineWeightedHemispherePdf(n, l);
        return F3(dot(n, l) / PI);
    }
    if (env.flags & F_MATERIAL_MIS) {
        is_transmission = false;
        use_mis = true;
        LayerProbs p = LayerProbabilities(sp, v);
        BsdfLayer layer = SelectBsdf(u.x, p);
        float2 u2 = {u.y, u.z};
        switch (layer) {
            case LAYER_ALPHA:
                l = -v; use_mis = false; pdf = p.alpha; is_transmission = true;
                return F3(1 - sp.alpha);
            case LAYER_DIFFUSE: l = SampleCosineWeightedHemisphere(sp.shading_normal, u2); break;   // :462-465
            case LAYER_SPECULAR: l = SampleSpecular(sp, v, u2); break;
            case LAYER_SHEEN: l = SampleCosineWeightedHemisphere(sp.shading_normal, u2); break;     // :418-421
            case LAYER_CLEARCOAT: l = SampleClearcoat(sp, v, u2); break;
            case LAYER_TRANSMISSION: l = SampleTransmission(sp, v, u2); is_transmission = true; break;
        }
        pdf = BsdfPdf(sp, v, l, is_transmission, p);
        return sp.alpha * GltfBsdf(*env.lut, sp, v, l, true, is_transmission);
    }
    if (u.x > sp.alpha) {
        l = -v; use_mis = false; pdf = (1 - sp.alpha); is_transmission = true;
        return F3(1 - sp.alpha);
    }
    use_mis = true;
    is_transmission = false;
    float3 n = sp.shading_normal;
    l = SampleCosineWeightedHemisphere(n, {u.y, u.z});
    pdf = CosineWeightedHemispherePdf(n, l);
    pdf *= sp.alpha;
    return sp.alpha * GltfBsdf(*env.lut, sp, v, l, false, false);
}
inline bool RussianRoulette(float mn, float mx, float u, float3& throughput, float3& weight) {   // :712-722
    float p = MaxValue(throughput);
    p = clamp(p, mn, mx);
    if (u < p) { weight /= p; return true; }
    return false;
}
// PathTracer.lib.hlsl:260-268 (Ray Tracing Gems ch. 6)
inline float3 OffsetRay(float3 p, float3 ng) {
    const float origin = 1.0f / 32.0f, float_scale = 1.0f / 65536.0f, int_scale = 256.0f;
    int of[3] = {f2i(int_scale * ng.x), f2i(int_scale * ng.y), f2i(int_scale * ng.z)};
    float pp[3] = {p.x, p.y, p.z}, nn[3] = {ng.x, ng.y, ng.z}, out[3];
    for (int i = 0; i < 3; i++) {
        float pi = asfloat(asint(pp[i]) + (pp[i] < 0 ? -of[i] : of[i]));
        out[i] = fabsf(pp[i]) < origin ? pp[i] + float_scale * nn[i] : pi;
    }
    return {out[0], out[1], out[2]};
}
// PathTracer.lib.hlsl:306-316
inline float3 NormalAdaptation(float3 ng, float3 ns, float3 v) {
    float3 r = reflect(-v, ns);
    float rdng = dot(r, ng);
    if (rdng < 0) return normalize(v + normalize(r - rdng * ng));
    return ns;
}

}  // namespace orc
