// oracle/hlsl.h -- TEST INFRASTRUCTURE ONLY (CPU oracle).  Not part of the product: nothing
// under gltf_renderer_amd/ may include, link or call this.
//
// Minimal HLSL-semantics vector layer for the CPU restatement of the reference shaders
// (SURVEY.md section 10 cheat-sheet).  Scalar fp32 everywhere; compile without fast-math and
// with -ffp-contract=off so the arithmetic is what is written.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace hlsl {

static const float PI = 3.14159265359f;   // Common.hlsli:8
static const float TAU = 2 * PI;          // Common.hlsli:9

struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
struct int2 { int x, y; };

inline float2 F2(float a) { return {a, a}; }
inline float3 F3(float a) { return {a, a, a}; }
inline float4 F4(float a) { return {a, a, a, a}; }
inline float4 F4(float3 v, float w) { return {v.x, v.y, v.z, w}; }
inline float3 xyz(float4 v) { return {v.x, v.y, v.z}; }
inline float2 xy(float3 v) { return {v.x, v.y}; }
inline float2 xy(float4 v) { return {v.x, v.y}; }

#define HLSL_OP2(T, op)                                                                      \
    inline T operator op(T a, T b);                                                          \
    inline T operator op(T a, float b);                                                      \
    inline T operator op(float a, T b);
inline float2 operator+(float2 a, float2 b) { return {a.x + b.x, a.y + b.y}; }
inline float2 operator-(float2 a, float2 b) { return {a.x - b.x, a.y - b.y}; }
inline float2 operator*(float2 a, float2 b) { return {a.x * b.x, a.y * b.y}; }
inline float2 operator/(float2 a, float2 b) { return {a.x / b.x, a.y / b.y}; }
inline float2 operator+(float2 a, float b) { return {a.x + b, a.y + b}; }
inline float2 operator-(float2 a, float b) { return {a.x - b, a.y - b}; }
inline float2 operator*(float2 a, float b) { return {a.x * b, a.y * b}; }
inline float2 operator/(float2 a, float b) { return {a.x / b, a.y / b}; }
inline float2 operator*(float a, float2 b) { return {a * b.x, a * b.y}; }
inline float2 operator-(float2 a) { return {-a.x, -a.y}; }

inline float3 operator+(float3 a, float3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline float3 operator-(float3 a, float3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float3 operator*(float3 a, float3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline float3 operator/(float3 a, float3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline float3 operator+(float3 a, float b) { return {a.x + b, a.y + b, a.z + b}; }
inline float3 operator-(float3 a, float b) { return {a.x - b, a.y - b, a.z - b}; }
inline float3 operator*(float3 a, float b) { return {a.x * b, a.y * b, a.z * b}; }
inline float3 operator/(float3 a, float b) { return {a.x / b, a.y / b, a.z / b}; }
inline float3 operator+(float a, float3 b) { return {a + b.x, a + b.y, a + b.z}; }
inline float3 operator-(float a, float3 b) { return {a - b.x, a - b.y, a - b.z}; }
inline float3 operator*(float a, float3 b) { return {a * b.x, a * b.y, a * b.z}; }
inline float3 operator-(float3 a) { return {-a.x, -a.y, -a.z}; }
inline float3& operator+=(float3& a, float3 b) { a = a + b; return a; }
inline float3& operator*=(float3& a, float3 b) { a = a * b; return a; }
inline float3& operator*=(float3& a, float b) { a = a * b; return a; }
inline float3& operator/=(float3& a, float b) { a = a / b; return a; }

inline float4 operator+(float4 a, float4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline float4 operator-(float4 a, float4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
inline float4 operator*(float4 a, float4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
inline float4 operator*(float4 a, float b) { return {a.x * b, a.y * b, a.z * b, a.w * b}; }
inline float4 operator*(float a, float4 b) { return {a * b.x, a * b.y, a * b.z, a * b.w}; }
inline float4 operator/(float4 a, float b) { return {a.x / b, a.y / b, a.z / b, a.w / b}; }
inline float4 operator-(float4 a) { return {-a.x, -a.y, -a.z, -a.w}; }

inline float dot(float2 a, float2 b) { return a.x * b.x + a.y * b.y; }
inline float dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float dot(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
inline float3 cross(float3 a, float3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline float length(float2 v) { return sqrtf(dot(v, v)); }
inline float length(float3 v) { return sqrtf(dot(v, v)); }
// normalize(v) = v * rsqrt(dot(v,v)): NaN for the zero vector (scrubbed per sample later).
inline float2 normalize(float2 v) { return v / sqrtf(dot(v, v)); }
inline float3 normalize(float3 v) { return v / sqrtf(dot(v, v)); }
inline float4 normalize(float4 v) { return v / sqrtf(dot(v, v)); }

// HLSL min/max return the non-NaN operand: exactly fminf/fmaxf.
inline float hmin(float a, float b) { return fminf(a, b); }
inline float hmax(float a, float b) { return fmaxf(a, b); }
inline float2 hmax(float2 a, float2 b) { return {hmax(a.x, b.x), hmax(a.y, b.y)}; }
inline float3 hmin(float3 a, float3 b) { return {hmin(a.x, b.x), hmin(a.y, b.y), hmin(a.z, b.z)}; }
inline float3 hmax(float3 a, float3 b) { return {hmax(a.x, b.x), hmax(a.y, b.y), hmax(a.z, b.z)}; }
inline float clamp(float x, float a, float b) { return hmin(hmax(x, a), b); }
inline float3 clamp(float3 v, float a, float b) { return {clamp(v.x, a, b), clamp(v.y, a, b), clamp(v.z, a, b)}; }
inline float saturate(float x) { return hmin(hmax(x, 0.0f), 1.0f); }   // NaN -> 0
inline float3 saturate(float3 v) { return {saturate(v.x), saturate(v.y), saturate(v.z)}; }
inline float lerp(float a, float b, float t) { return a + t * (b - a); }
inline float3 lerp(float3 a, float3 b, float t) { return a + t * (b - a); }
inline float3 lerp(float3 a, float3 b, float3 t) { return a + t * (b - a); }
inline float4 lerp(float4 a, float4 b, float t) { return a + t * (b - a); }
inline float sign(float x) { return x > 0 ? 1.0f : (x < 0 ? -1.0f : 0.0f); }  // sign(0) = 0
inline float3 reflect(float3 i, float3 n) { return i - 2 * dot(n, i) * n; }
// sin / cos: HLSL leaves their precision to the implementation.  The oracle DEFINES them as the correctly rounded value -- evaluated in
// double, rounded once -- so that its images do not depend on which float routine (and which FMA variant of it) the host's libm picks;
// the host's sinf / cosf are within 0.56 ulp of that but not always equal to it, and a last bit in a sampled direction is a different path
// on a scene that amplifies rounding.  The HIP path evaluates them the same way (pt_math.h pt_sincos).
inline float o_sin(float x) { return (float)std::sin((double)x); }
inline float o_cos(float x) { return (float)std::cos((double)x); }
// atan2, log2, exp2, and pow / exp through them: DEFINED as float kernels (IEEE multiplies, adds, divisions in a fixed order; this file is
// compiled without contraction), which the HIP path states operation for operation (csrc/pt_math.h co_atan2 / co_log2 / co_exp2): the same
// bits on both sides.  HLSL leaves their precision open; these are within 1.3 / 2.9 / 1.2 ulp (tools/fit_transcendentals.py; checked
// against libm in tests/test_oracle_kat.py).  The float routines of two libraries (glibc here, v_exp_f32 / v_log_f32 / ocml there) differ
// in the last bit in several percent of their results, and double evaluation is slow on the GPU.
inline uint32_t o_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float o_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
inline float o_atan2(float y, float x) {
    if (!(x == x) || !(y == y)) return NAN;
    const float ax = fabsf(x), ay = fabsf(y), mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    float a;
    if (mx == 0.0f) a = 0.0f;
    else if (mx == INFINITY) a = mn == INFINITY ? 1.0f : 0.0f;
    else a = mn / mx;
    const float s = a * a;
    float q = -0x1.40bebap-9f;
    q = q * s + 0x1.c293dap-7f; q = q * s + -0x1.2920fep-5f; q = q * s + 0x1.0168bap-4f; q = q * s + -0x1.634104p-4f;
    q = q * s + 0x1.c41dd4p-4f; q = q * s + -0x1.246facp-3f; q = q * s + 0x1.999860p-3f; q = q * s + -0x1.555554p-2f;
    float r = a + a * (s * q);
    if (ay > ax) r = 1.57079637f - r;
    if (o_bits(x) >> 31) r = 3.14159274f - r;
    return copysignf(r, y);
}
inline float o_log2(float x) {
    if (!(x > 0.0f)) return x == 0.0f ? -INFINITY : NAN;
    if (x == INFINITY) return x;
    int e;
    float m = frexpf(x, &e);                                 // [1/2, 1), exact
    if (m < 0.707106769f) { m *= 2.0f; e -= 1; }
    const float t = (m - 1.0f) / (m + 1.0f), s = t * t;
    float q = 0x1.ba1838p-2f;
    q = q * s + 0x1.274720p-1f; q = q * s + 0x1.ec70e6p-1f; q = q * s + 0x1.715476p+1f;
    return (float)e + t * q;
}
inline float o_exp2_reduced(float r) {                       // 2^r, r in [-1/2, 1/2]
    float q = 0x1.444004p-13f;
    q = q * r + 0x1.5f0896p-10f; q = q * r + 0x1.3b2a1cp-7f; q = q * r + 0x1.c6af6cp-5f; q = q * r + 0x1.ebfbe0p-3f; q = q * r + 0x1.62e430p-1f;
    return 1.0f + r * q;
}
inline float o_scale2(float v, float n) { return ldexpf(v, (int)n); }                        // v * 2^n, n an integer in [-125, 128]: exact, or +inf
inline float o_exp2(float p) {
    if (!(p == p)) return p;
    if (p >= 128.0f) return INFINITY;
    if (p < -125.0f) return 0.0f;                            // results below the normal range are zero
    const float n = rintf(p);
    return o_scale2(o_exp2_reduced(p - n), n);                // (p - n is exact)
}
// e^x: n = round(x / ln 2), r = x - n ln 2 with ln 2 in two parts (the first has 11 trailing zero bits: n times it is exact), e^r = 2^(r / ln 2)
inline float o_exp(float x) {
    if (!(x == x)) return x;
    if (x > 88.75f) return INFINITY;
    if (x < -86.5f) return 0.0f;
    const float n = rintf(x * 1.44269504f);
    const float r = (x - n * 0.693145751953125f) - n * 1.42860677e-06f;
    return o_scale2(o_exp2_reduced(r * 1.44269504f), n);
}
// pow(x,y) = exp2(y*log2(x)): x<0 -> NaN, pow(0, y>0) = 0.
inline float hpow(float x, float y) { return o_exp2(y * o_log2(x)); }
// pow with the constant integer exponents 5 (Schlick's Fresnel, Bsdf.hlsli:39-47) and 4 (the light falloff, Lights.hlsli:41) as correctly rounded
// products, NaN for a negative base like the exp2 / log2 form: HLSL leaves pow's precision open (and shader compilers expand such pows); this is
// the definition the HIP kernels evaluate to the same bits (csrc/pt_math.h hpow5 / hpow4) -- two approximate exp2 / log2 libraries are not.
inline float hpow5(float x) { const float x2 = x * x; return x < 0.0f ? NAN : (x2 * x2) * x; }
inline float hpow4(float x) { const float x2 = x * x; return x < 0.0f ? NAN : x2 * x2; }
inline float3 hpow(float3 v, float y) { return {hpow(v.x, y), hpow(v.y, y), hpow(v.z, y)}; }
inline float3 habs(float3 v) { return {fabsf(v.x), fabsf(v.y), fabsf(v.z)}; }
inline bool any_gt0(float3 v) { return v.x > 0 || v.y > 0 || v.z > 0; }
inline bool any_nan(float3 v) { return std::isnan(v.x) || std::isnan(v.y) || std::isnan(v.z); }
inline bool any_inf(float3 v) { return std::isinf(v.x) || std::isinf(v.y) || std::isinf(v.z); }
inline int asint(float f) { int i; memcpy(&i, &f, 4); return i; }
inline float asfloat(int i) { float f; memcpy(&f, &i, 4); return f; }
// (int)(float): truncate toward zero; keep it defined for NaN / out of range.
inline int f2i(float f) {
    if (!(f == f)) return 0;
    if (f >= 2147483520.0f) return 2147483647;
    if (f <= -2147483648.0f) return (int)0x80000000;
    return (int)f;
}
// (uint)(float): negative / NaN -> 0.
inline uint32_t f2u(float f) {
    if (!(f > 0)) return 0;
    if (f >= 4294967040.0f) return 0xffffffffu;
    return (uint32_t)f;
}

// float4x4 as stored by glm (column-major, m[col*4+row]); mul(M, v) = ordinary M*v.
struct float4x4 { float m[16]; };
inline float4 mul(const float4x4& M, float4 v) {
    float4 r;
    r.x = M.m[0] * v.x + M.m[4] * v.y + M.m[8] * v.z + M.m[12] * v.w;
    r.y = M.m[1] * v.x + M.m[5] * v.y + M.m[9] * v.z + M.m[13] * v.w;
    r.z = M.m[2] * v.x + M.m[6] * v.y + M.m[10] * v.z + M.m[14] * v.w;
    r.w = M.m[3] * v.x + M.m[7] * v.y + M.m[11] * v.z + M.m[15] * v.w;
    return r;
}
// float3x3(a,b,c) has ROWS a,b,c; mul(M,v) = (a.v, b.v, c.v).
struct float3x3 { float3 r0, r1, r2; };
inline float3x3 M3(float3 a, float3 b, float3 c) { return {a, b, c}; }
inline float3 mul(const float3x3& M, float3 v) { return {dot(M.r0, v), dot(M.r1, v), dot(M.r2, v)}; }
inline float3x3 transpose(const float3x3& M) {
    return {{M.r0.x, M.r1.x, M.r2.x}, {M.r0.y, M.r1.y, M.r2.y}, {M.r0.z, M.r1.z, M.r2.z}};
}
inline float3x3 mul(const float3x3& A, const float3x3& B) {
    float3x3 Bt = transpose(B);
    return {{dot(A.r0, Bt.r0), dot(A.r0, Bt.r1), dot(A.r0, Bt.r2)},
            {dot(A.r1, Bt.r0), dot(A.r1, Bt.r1), dot(A.r1, Bt.r2)},
            {dot(A.r2, Bt.r0), dot(A.r2, Bt.r1), dot(A.r2, Bt.r2)}};
}

// IEEE binary16 <-> binary32 (round to nearest even; overflow -> inf, as R16G16B16A16_FLOAT).
inline float half_to_float(uint16_t h) {
    uint32_t s = (h >> 15) & 1, e = (h >> 10) & 0x1f, m = h & 0x3ff, out;
    if (e == 0) {
        if (m == 0) out = s << 31;
        else {
            int sh = 0;
            while (!(m & 0x400)) { m <<= 1; sh++; }
            m &= 0x3ff;
            out = (s << 31) | ((uint32_t)(127 - 15 - sh + 1) << 23) | (m << 13);
        }
    } else if (e == 31) out = (s << 31) | 0x7f800000u | (m << 13);
    else out = (s << 31) | ((e + 112) << 23) | (m << 13);
    float f; memcpy(&f, &out, 4); return f;
}
inline uint16_t float_to_half(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    uint32_t s = (x >> 16) & 0x8000u; x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(s | 0x7c00u | ((x > 0x7f800000u) ? 0x200u : 0));
    if (x >= 0x477ff000u) return (uint16_t)(s | 0x7c00u);               // >= 65520 -> inf
    if (x < 0x33000001u) return (uint16_t)s;                              // < 2^-25 -> 0
    int e = (int)(x >> 23) - 127;
    uint32_t m = (x & 0x7fffffu) | 0x800000u;
    int shift = (e < -14) ? (13 + (-14 - e)) : 13;
    uint32_t hm = m >> shift, rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (hm & 1))) hm++;
    uint32_t he = (e < -14) ? 0 : (uint32_t)(e + 15);
    // hm holds the implicit bit for normals: adding it to (he-1)<<10 carries correctly.
    uint32_t out = (e < -14) ? hm : (((he - 1) << 10) + hm);
    return (uint16_t)(s | out);
}

}  // namespace hlsl
