// lean_div_probe.hip -- is a shorter fp32 division bit-identical to the compiler's IEEE sequence (2 x v_div_scale, v_rcp, 6 fma-class,
// v_div_fmas, v_div_fixup: ~48 issue cycles for a lone wave) on the operands a shader meets?  Random normal operands over a wide
// exponent range; counts mismatches of   L1: rcp + one Newton step + one quotient correction (28 cycles)   L2: + a second correction (36)
// and of the same for sqrt (v_sqrt + one / two corrections).     build: hipcc -O3 --offload-arch=gfx950 -o lean_div_probe lean_div_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include "../../gltf_renderer_amd/csrc/pt_math.h"      // pt::fdiv, pt::hpow: the library's own functions are what is checked
__device__ __forceinline__ uint32_t pcg(uint32_t& s) { s = s * 747796405u + 2891336453u; uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u; return (w >> 22u) ^ w; }
__device__ __forceinline__ float rnd_float(uint32_t& s, int emin, int emax) {          // random sign, exponent in [emin, emax], random mantissa
    const uint32_t m = pcg(s) & 0x7fffffu, e = (uint32_t)(127 + emin) + pcg(s) % (uint32_t)(emax - emin + 1), sg = pcg(s) & 0x80000000u;
    return __uint_as_float(sg | (e << 23) | m);
}
__device__ __forceinline__ float div_l1(float a, float b) { return pt::fdiv(a, b); }     // L1 = pt_math.h's fdiv
__device__ __forceinline__ float div_l2(float a, float b) {
    float r = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float q = a * r;
    float m = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(m, r, q);
    m = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(m, r, q);
}
__device__ __forceinline__ float sqrt_l1(float x) {            // g ~ sqrt(x), h ~ 1/(2 sqrt(x)); one Goldschmidt-style correction of g
    const float g0 = __builtin_amdgcn_sqrtf(x);
    const float h = 0.5f * __builtin_amdgcn_rcpf(g0);
    const float d = __builtin_fmaf(-g0, g0, x);
    return __builtin_fmaf(d, h, g0);
}
__device__ __forceinline__ float sqrt_l2(float x) {
    float g = sqrt_l1(x);
    const float h = 0.5f * __builtin_amdgcn_rcpf(g);
    const float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}
__global__ void k(unsigned long long* bad, int rounds, int emin, int emax) {
    uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 99u;
    unsigned b1 = 0, b2 = 0, s1 = 0, s2 = 0;
    for (int i = 0; i < rounds; i++) {
        const float a = rnd_float(s, emin, emax), b = rnd_float(s, emin, emax);
        const float ref = a / b;
        b1 += __float_as_uint(div_l1(a, b)) != __float_as_uint(ref);
        b2 += __float_as_uint(div_l2(a, b)) != __float_as_uint(ref);
        const float x = fabsf(a), sr = sqrtf(x);
        s1 += __float_as_uint(sqrt_l1(x)) != __float_as_uint(sr);
        s2 += __float_as_uint(sqrt_l2(x)) != __float_as_uint(sr);
    }
    atomicAdd(&bad[0], b1); atomicAdd(&bad[1], b2); atomicAdd(&bad[2], s1); atomicAdd(&bad[3], s2);
}
// The operands OUTSIDE fdiv's contract: what it returns where the IEEE quotient is +-inf or a large finite number.
__global__ void k_special(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ lean, float* __restrict__ ieee, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { lean[i] = pt::fdiv(a[i], b[i]); ieee[i] = a[i] / b[i]; }
}
static void special_cases() {
    const float A[] = {3.0e38f, 1.0e30f, -2.0e20f, 1.0f, 5.0f, 1.0e-30f, 1.0f, 0.0f, 1.0f, INFINITY};
    const float B[] = {1.0e-3f, 1.0e-20f, 1.0e-25f, 1.0e-40f, -3.0e-42f, 1.0e-39f, 0.0f, 0.0f, INFINITY, 2.0f};
    const int n = (int)(sizeof(A) / sizeof(A[0]));
    float *da, *db, *dl, *di, L[16], I[16];
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dl, n * 4); hipMalloc(&di, n * 4);
    hipMemcpy(da, A, n * 4, hipMemcpyHostToDevice); hipMemcpy(db, B, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_special, dim3(1), dim3(64), 0, 0, da, db, dl, di, n);
    hipMemcpy(L, dl, n * 4, hipMemcpyDeviceToHost); hipMemcpy(I, di, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; i++) printf("special %d: %g / %g -> fdiv %g , ieee %g\n", i, (double)A[i], (double)B[i], (double)L[i], (double)I[i]);
}
int main() {
    special_cases();
    unsigned long long* d; hipMalloc(&d, 32);
    const int ranges[][2] = {{-10, 10}, {-40, 40}, {-60, 60}, {-1, 0}};
    for (auto& r : ranges) {
        hipMemset(d, 0, 32);
        const int blocks = 4096, threads = 256, rounds = 2000;
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, rounds, r[0], r[1]);
        unsigned long long h[4]; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
        printf("exponents [%d, %d], %.2e operand pairs: division mismatches L1 %llu  L2 %llu ; sqrt mismatches L1 %llu  L2 %llu\n", r[0], r[1], (double)blocks * threads * rounds, h[0], h[1], h[2], h[3]);
    }
    return 0;
}
