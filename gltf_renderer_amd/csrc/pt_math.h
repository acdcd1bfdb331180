// pt_math.h -- device vector math with HLSL semantics for the gfx950 path-tracing kernels.
// (saturate NaN->0, sign(0)=0, pow = exp2(y*log2 x), normalize(0)=NaN,
//  min/max return the non-NaN operand.)
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#define PT_DEV __device__ __forceinline__

namespace pt {

constexpr float kPi = 3.14159265359f;      // Common.hlsli:8
constexpr float kTau = 2.0f * kPi;         // Common.hlsli:9

struct vec2 { float x, y; };
struct vec3 { float x, y, z; };
struct vec4 { float x, y, z, w; };

// Loads through a pointer that was itself read from memory (stream pointers in instance rows, texel pointers in material
// slots) would compile to flat_load, which ties up both the vector-memory and the LDS wait counters.  Everything the kernels
// gather lives in HBM, so say so: these compile to global_load.
#define PT_GLOBAL __attribute__((address_space(1)))
typedef float pt_f4n __attribute__((ext_vector_type(4)));
typedef float pt_f2n __attribute__((ext_vector_type(2)));
typedef uint32_t pt_u2n __attribute__((ext_vector_type(2)));
typedef uint32_t pt_u4n __attribute__((ext_vector_type(4)));
template <class T> PT_DEV T gload(const T* p) { return *(const PT_GLOBAL T*)p; }                     // scalar types only
PT_DEV float4 gload_f4(const void* p) { pt_f4n v = *(const PT_GLOBAL pt_f4n*)p; return make_float4(v.x, v.y, v.z, v.w); }
PT_DEV float2 gload_f2(const void* p) { pt_f2n v = *(const PT_GLOBAL pt_f2n*)p; return make_float2(v.x, v.y); }
PT_DEV uint2 gload_u2(const void* p) { pt_u2n v = *(const PT_GLOBAL pt_u2n*)p; return make_uint2(v.x, v.y); }
PT_DEV uint4 gload_u4(const void* p) { pt_u4n v = *(const PT_GLOBAL pt_u4n*)p; return make_uint4(v.x, v.y, v.z, v.w); }

// x / C for an integer-valued x in [0, 65535] and C = 255, 1023 or 65535 (unorm decoding): three instructions that give the
// correctly rounded quotient -- bit for bit what the IEEE division gives, which costs ten.  q = x * RN(1/C) is within an ulp,
// fma(-C, q, x) is its exact remainder, one more fma corrects it (Markstein).  Exhaustively checked for every x of the three
// ranges against `x / C` (tests/test_host_abi.py).
template <int C> PT_DEV float unorm_div(float x) {
    constexpr float inv = 1.0f / (float)C;
    const float q = x * inv;
    return __builtin_fmaf(__builtin_fmaf(-(float)C, q, x), inv, q);
}

PT_DEV vec3 v3(float a) { return {a, a, a}; }
PT_DEV vec3 v3(float x, float y, float z) { return {x, y, z}; }
PT_DEV vec3 v3p(const float* p) { return {p[0], p[1], p[2]}; }
PT_DEV vec3 xyz(vec4 v) { return {v.x, v.y, v.z}; }

// fp32 division.  The compiler's IEEE sequence is eleven instructions (2 x v_div_scale, v_rcp, six fma-class, v_div_fmas, v_div_fixup:
// ~48 issue cycles of a lone wave, the transcendental counts double) and a hit's shading runs ~150 of them -- 11 % of the shade stage
// (measured with -fno-hip-fp32-correctly-rounded-divide-sqrt, which is NOT used: its 2.5-ulp quotients doubled the share of pixel-samples
// that part from the oracle and broke the exact white furnace).  fdiv() is the same Newton-Raphson core without the operand scaling:
// rcp, one correction of the reciprocal, one of the quotient, then v_div_fixup for the special operands (zero, infinity, NaN).  It is
// BIT-IDENTICAL to a / b whenever divisor, dividend and quotient are normal numbers (tools/probes/lean_div_probe.hip: 0 mismatches in
// 8.4e9 random operand pairs over exponents -60..60).  OUTSIDE that contract it is not IEEE division, because nothing rescales the operands: a
// quotient below 2^-126 is not correctly rounded, a SUBNORMAL divisor (v_rcp treats it as zero: the reciprocal is infinite) returns +-inf
// where a / b is a large finite number (1e-30 / 1e-39: inf for 1e9), and a divisor above 2^126 (its reciprocal is subnormal: zero) returns 0.  A quotient that overflows comes out +-inf like a / b, and zero, infinite
// and NaN operands are repaired by v_div_fixup (measured on the MI355X: tools/probes/lean_div_probe.hip prints the cases,
// tests/test_gpu_math.py pins them).  On the path that reaches contrib / light_pdf, the MIS ratios and the luminance clamp a subnormal pdf is
// already a degenerate sample that sanitize_sample zeroes either way (NaN and Inf alike, PathTracer.lib.hlsl:760-766).  Where such operands
// can occur by construction (1 / direction of the ray set-up) the code keeps the compiler's sequence.
// A vector divided by one scalar corrects the reciprocal once.  PT_EXACT_DIV=1 restores the compiler's sequence everywhere (A/B).
#ifndef PT_EXACT_DIV
#define PT_EXACT_DIV 0
#endif
PT_DEV float frcp_refined(float b) {                 // 1 / b to within half an ulp (not always the rounded reciprocal; fdiv_with corrects the quotient)
    float r = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
PT_DEV float fdiv_with(float a, float b, float r) {  // a / b given r = frcp_refined(b), or the rounded 1 / b of a constant b
#if PT_EXACT_DIV
    return a / b;
#else
    const float q = a * r;
    const float m = __builtin_fmaf(-b, q, a);
    return __builtin_amdgcn_div_fixupf(__builtin_fmaf(m, r, q), b, a);
#endif
}
PT_DEV float fdiv(float a, float b) {
#if PT_EXACT_DIV
    return a / b;
#else
    return fdiv_with(a, b, frcp_refined(b));
#endif
}

PT_DEV vec2 operator+(vec2 a, vec2 b) { return {a.x + b.x, a.y + b.y}; }
PT_DEV vec2 operator-(vec2 a, vec2 b) { return {a.x - b.x, a.y - b.y}; }
PT_DEV vec2 operator*(vec2 a, float b) { return {a.x * b, a.y * b}; }
PT_DEV vec2 operator*(float a, vec2 b) { return {a * b.x, a * b.y}; }

PT_DEV vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
PT_DEV vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
PT_DEV vec3 operator*(vec3 a, vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
PT_DEV vec3 operator/(vec3 a, vec3 b) { return {fdiv(a.x, b.x), fdiv(a.y, b.y), fdiv(a.z, b.z)}; }
PT_DEV vec3 operator+(vec3 a, float b) { return {a.x + b, a.y + b, a.z + b}; }
PT_DEV vec3 operator-(vec3 a, float b) { return {a.x - b, a.y - b, a.z - b}; }
PT_DEV vec3 operator*(vec3 a, float b) { return {a.x * b, a.y * b, a.z * b}; }
PT_DEV vec3 operator/(vec3 a, float b) { const float r = frcp_refined(b); return {fdiv_with(a.x, b, r), fdiv_with(a.y, b, r), fdiv_with(a.z, b, r)}; }
PT_DEV vec3 operator*(float a, vec3 b) { return {a * b.x, a * b.y, a * b.z}; }
PT_DEV vec3 operator+(float a, vec3 b) { return {a + b.x, a + b.y, a + b.z}; }
PT_DEV vec3 operator-(float a, vec3 b) { return {a - b.x, a - b.y, a - b.z}; }
PT_DEV vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }
PT_DEV vec3& operator+=(vec3& a, vec3 b) { a = a + b; return a; }
PT_DEV vec3& operator*=(vec3& a, vec3 b) { a = a * b; return a; }
PT_DEV vec3& operator*=(vec3& a, float b) { a = a * b; return a; }
PT_DEV vec4 operator+(vec4 a, vec4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
PT_DEV vec4 operator*(vec4 a, vec4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
PT_DEV vec4 operator*(vec4 a, float b) { return {a.x * b, a.y * b, a.z * b, a.w * b}; }

PT_DEV float dot(vec2 a, vec2 b) { return a.x * b.x + a.y * b.y; }
PT_DEV float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PT_DEV vec3 cross(vec3 a, vec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
PT_DEV float length(vec3 v) { return sqrtf(dot(v, v)); }
PT_DEV vec3 normalize(vec3 v) { return v / sqrtf(dot(v, v)); }
PT_DEV vec2 normalize(vec2 v) { const float l = sqrtf(dot(v, v)), r = frcp_refined(l); return {fdiv_with(v.x, l, r), fdiv_with(v.y, l, r)}; }

PT_DEV float hmin(float a, float b) { return fminf(a, b); }
PT_DEV float hmax(float a, float b) { return fmaxf(a, b); }
PT_DEV vec3 hmin(vec3 a, vec3 b) { return {fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)}; }
PT_DEV vec3 hmax(vec3 a, vec3 b) { return {fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)}; }
PT_DEV float clampf(float x, float a, float b) { return fminf(fmaxf(x, a), b); }
PT_DEV float saturate(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
PT_DEV float lerpf(float a, float b, float t) { return a + t * (b - a); }
PT_DEV vec3 lerp3(vec3 a, vec3 b, float t) { return a + t * (b - a); }
PT_DEV float signf(float x) { return x > 0 ? 1.0f : (x < 0 ? -1.0f : 0.0f); }
PT_DEV vec3 reflect(vec3 i, vec3 n) { return i - 2 * dot(n, i) * n; }
// sin and cos are evaluated in DOUBLE precision and rounded once: the correctly rounded fp32 value in all but one case in ~1e9, which is
// what the CPU oracle computes the same way (its libm's double sin / cos; tests/test_gpu_round3.py: bit-identical on 10^6 arguments).  The
// device library's fp32 sinf / cosf are within 1-2 ulp of that -- and on the Sponza-class scene those last bits, in the directions the BSDF
// samplers and the environment sampler draw, made 0.04 % of the pixel-samples take another path than the oracle's (round 2: 8.4e-4 at
// 64 spp with them, 4.9e-3 without).  The cost is not measurable: the hardware's approximate v_sin_f32 / v_cos_f32 in their place
// (PT_PROBE_FAST_SINCOS, wrong images) save 0.05 ms of a 23-ms launch -- two calls a hit.  The other transcendentals (atan2, log2, exp2,
// exp, pow) are float kernels defined below: through double THEY are expensive (7.7 ms a launch).
// PT_F64_TRANSCENDENTALS=0: the fp32 library sinf / cosf (A/B).
#ifndef PT_F64_TRANSCENDENTALS
#define PT_F64_TRANSCENDENTALS 1
#endif
#if PT_F64_TRANSCENDENTALS
PT_DEV float pt_sin(float x) { return (float)sin((double)x); }
PT_DEV float pt_cos(float x) { return (float)cos((double)x); }
#else
PT_DEV float pt_sin(float x) { return sinf(x); }
PT_DEV float pt_cos(float x) { return cosf(x); }
#endif
// both of one angle: one argument reduction in double for the pair (PT_SINCOS_PAIR=0: two separate calls, for A/B)
#ifndef PT_SINCOS_PAIR
#define PT_SINCOS_PAIR 1
#endif
PT_DEV void pt_sincos(float x, float& s, float& c) {
#ifdef PT_PROBE_FAST_SINCOS      // PROBE ONLY: the hardware's approximate v_sin_f32 / v_cos_f32 -- what an ideal sincos would save
    s = __sinf(x); c = __cosf(x); return;
#endif
#if PT_F64_TRANSCENDENTALS && PT_SINCOS_PAIR
    double ds, dc;
    sincos((double)x, &ds, &dc);
    s = (float)ds; c = (float)dc;
#else
    s = pt_sin(x); c = pt_cos(x);
#endif
}
// atan2, log2, exp2 -- and through them pow and exp -- are DEFINED here, as short float kernels made of IEEE multiplies, adds and divisions in
// a fixed order (no fused multiply-add: these translation units are compiled without contraction), which the CPU oracle states operation for
// operation (its hlsl.h): the same bits on both sides by construction.  HLSL leaves the precision of these intrinsics to the implementation;
// these are within 1.3 (atan), 2.9 (log2) and 1.2 (exp2) ulp (tools/fit_transcendentals.py fits and measures them).  Why not the library's:
// v_exp_f32 / v_log_f32 / ocml's atan2f and glibc's routines differ in the last bit in several percent of their results.  atan2 sits in the
// direction -> importance-map texel mapping of the environment pdf (a last bit picks the neighbouring texel once in ~10^6 lookups: a MIS
// weight off by up to a percent), pow and exp in the sheen lobe (five pows and two exps per evaluation).  Through double (like sin / cos)
// they cost 7.7 ms of a 22.6 ms launch (measured: ocml's double atan2, log2, exp2 are long).  These: atan2 nothing measurable; pow / exp
// 0.35 ms on the Sponza-class scene, whose curtains (3 % of the hits) put a sheen lane into most waves of the shade stage -- v_log_f32 /
// v_exp_f32 were 3 instructions per pow, this is ~55 (0.8 ms before the view-dependent half of the sheen lobe was hoisted out of the three
// evaluations a hit makes, pt_shading.h prepare_sheen; as real calls instead of inlined code 1.0 ms).  PT_CO_TRANSCENDENTALS / PT_CO_ATAN2 /
// PT_CO_POW = 0: the library routines (A/B).
#ifndef PT_CO_TRANSCENDENTALS
#define PT_CO_TRANSCENDENTALS 1
#endif
PT_DEV float co_atan2(float y, float x) {
    if (!(x == x) || !(y == y)) return __builtin_nanf("");
    const float ax = fabsf(x), ay = fabsf(y), mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    float a;
    if (mx == 0.0f) a = 0.0f;
    else if (mx == __builtin_inff()) a = mn == __builtin_inff() ? 1.0f : 0.0f;
    else a = mn / mx;                                        // the compiler's correctly rounded division
    const float s = a * a;
    float q = -0x1.40bebap-9f;
    q = q * s + 0x1.c293dap-7f; q = q * s + -0x1.2920fep-5f; q = q * s + 0x1.0168bap-4f; q = q * s + -0x1.634104p-4f;
    q = q * s + 0x1.c41dd4p-4f; q = q * s + -0x1.246facp-3f; q = q * s + 0x1.999860p-3f; q = q * s + -0x1.555554p-2f;
    float r = a + a * (s * q);                               // atan(a), a in [0, 1]
    if (ay > ax) r = 1.57079637f - r;
    if (__float_as_uint(x) >> 31) r = 3.14159274f - r;
    return copysignf(r, y);
}
PT_DEV float co_log2(float x) {
    if (!(x > 0.0f)) return x == 0.0f ? -__builtin_inff() : __builtin_nanf("");
    if (x == __builtin_inff()) return x;
    float m = __builtin_amdgcn_frexp_mantf(x);               // x = m 2^e, m in [1/2, 1): exact, subnormal arguments included (frexpf in the oracle)
    int e = __builtin_amdgcn_frexp_expf(x);
    if (m < 0.707106769f) { m *= 2.0f; e -= 1; }             // [sqrt 1/2, sqrt 2)
    const float t = fdiv(m - 1.0f, m + 1.0f), s = t * t;     // (operands and quotient inside fdiv's contract: the IEEE quotient)
    float q = 0x1.ba1838p-2f;
    q = q * s + 0x1.274720p-1f; q = q * s + 0x1.ec70e6p-1f; q = q * s + 0x1.715476p+1f;
    return (float)e + t * q;
}
PT_DEV float co_exp2_reduced(float r) {                       // 2^r, r in [-1/2, 1/2]
    float q = 0x1.444004p-13f;
    q = q * r + 0x1.5f0896p-10f; q = q * r + 0x1.3b2a1cp-7f; q = q * r + 0x1.c6af6cp-5f; q = q * r + 0x1.ebfbe0p-3f; q = q * r + 0x1.62e430p-1f;
    return 1.0f + r * q;
}
PT_DEV float co_scale2(float v, float n) { return __builtin_amdgcn_ldexpf(v, (int)n); }     // v * 2^n, n an integer in [-125, 128]: exact, or +inf (ldexpf in the oracle)
PT_DEV float co_exp2(float p) {
    if (!(p == p)) return p;
    if (p >= 128.0f) return __builtin_inff();
    if (p < -125.0f) return 0.0f;                            // results below the normal range are zero
    const float n = rintf(p);
    return co_scale2(co_exp2_reduced(p - n), n);                // (p - n is exact)
}
// e^x: n = round(x / ln 2), r = x - n ln 2 with ln 2 in two parts (the first has 11 trailing zero bits: n times it is exact), e^r = 2^(r / ln 2)
PT_DEV float co_exp(float x) {
    if (!(x == x)) return x;
    if (x > 88.75f) return __builtin_inff();
    if (x < -86.5f) return 0.0f;
    const float n = rintf(x * 1.44269504f);
    const float r = (x - n * 0.693145751953125f) - n * 1.42860677e-06f;
    return co_scale2(co_exp2_reduced(r * 1.44269504f), n);
}
#ifndef PT_CO_ATAN2
#define PT_CO_ATAN2 PT_CO_TRANSCENDENTALS
#endif
#ifndef PT_CO_POW
#define PT_CO_POW PT_CO_TRANSCENDENTALS
#endif
#if PT_CO_ATAN2
PT_DEV float pt_atan2(float y, float x) { return co_atan2(y, x); }
#else
PT_DEV float pt_atan2(float y, float x) { return atan2f(y, x); }
#endif
// pow(x, y) = exp2(y * log2 x): NaN for a negative base, pow(0, y > 0) = 0.
#if PT_CO_POW
PT_DEV float pt_exp(float x) { return co_exp(x); }
PT_DEV float hpow(float x, float y) { return co_exp2(y * co_log2(x)); }
#else
PT_DEV float pt_exp(float x) { return expf(x); }
PT_DEV float hpow(float x, float y) { return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)); }
#endif
// pow with the CONSTANT integer exponents the path uses -- Schlick's (1 - |c|)^5, the punctual lights' (d / cutoff)^4 -- as products: x^2 * x^2 (* x),
// each product correctly rounded, NaN for a negative base like the exp2 / log2 form.  HLSL leaves pow's precision to the implementation and
// shader compilers expand such pows themselves; what matters here is that this is a definition the CPU oracle evaluates to the SAME BITS
// (its hlsl.h: hpow5 / hpow4), which exp2(y * log2 x) through two different approximate libraries is not: Schlick's weight feeds the lobe
// pick and Russian roulette, and a last-bit difference there flips about one pixel-sample in two million into a different path -- enough
// fireflies, at BASELINE size and 64 samples, to bring the image metric to 8e-4 of the 1e-3 contract (tools/fullsize_parity.py).
PT_DEV float hpow5(float x) { const float x2 = x * x; return x < 0.0f ? __builtin_nanf("") : (x2 * x2) * x; }
PT_DEV float hpow4(float x) { const float x2 = x * x; return x < 0.0f ? __builtin_nanf("") : x2 * x2; }
PT_DEV float max3(vec3 c) { return fmaxf(fmaxf(c.x, c.y), c.z); }
PT_DEV bool any_gt0(vec3 v) { return v.x > 0 || v.y > 0 || v.z > 0; }
PT_DEV bool any_nan(vec3 v) { return (v.x != v.x) || (v.y != v.y) || (v.z != v.z); }
PT_DEV bool any_inf(vec3 v) { return isinf(v.x) || isinf(v.y) || isinf(v.z); }
PT_DEV float heavyside(float a) { return a > 0 ? 1.f : 0.f; }
// (int)(float) / (uint)(float) with defined behaviour for NaN and overflow.
PT_DEV int f2i(float f) {
    if (!(f == f)) return 0;
    if (f >= 2147483520.0f) return 2147483647;
    if (f <= -2147483648.0f) return (int)0x80000000;
    return (int)f;
}
PT_DEV uint32_t f2u(float f) {
    if (!(f > 0)) return 0;
    if (f >= 4294967040.0f) return 0xffffffffu;
    return (uint32_t)f;
}
// column-major 4x4 (glm) times (v,1) / (v,0), xyz only
PT_DEV vec3 mul_point(const float* M, vec3 v) {
    return {M[0] * v.x + M[4] * v.y + M[8] * v.z + M[12], M[1] * v.x + M[5] * v.y + M[9] * v.z + M[13],
            M[2] * v.x + M[6] * v.y + M[10] * v.z + M[14]};
}
PT_DEV vec3 mul_dir(const float* M, vec3 v) {
    return {M[0] * v.x + M[4] * v.y + M[8] * v.z, M[1] * v.x + M[5] * v.y + M[9] * v.z, M[2] * v.x + M[6] * v.y + M[10] * v.z};
}
PT_DEV vec4 mul4(const float* M, vec4 v) {
    return {M[0] * v.x + M[4] * v.y + M[8] * v.z + M[12] * v.w, M[1] * v.x + M[5] * v.y + M[9] * v.z + M[13] * v.w,
            M[2] * v.x + M[6] * v.y + M[10] * v.z + M[14] * v.w, M[3] * v.x + M[7] * v.y + M[11] * v.z + M[15] * v.w};
}
// frame (t, b, n): to_local(v) = (t.v, b.v, n.v); to_world(h) = t*h.x + b*h.y + n*h.z
PT_DEV vec3 to_local(vec3 t, vec3 b, vec3 n, vec3 v) { return {dot(t, v), dot(b, v), dot(n, v)}; }
PT_DEV vec3 to_world(vec3 t, vec3 b, vec3 n, vec3 h) {
    return {t.x * h.x + b.x * h.y + n.x * h.z, t.y * h.x + b.y * h.y + n.y * h.z, t.z * h.x + b.z * h.y + n.z * h.z};
}
PT_DEV float half_bits_to_float(uint16_t h) { return __half2float(__ushort_as_half(h)); }

}  // namespace pt
