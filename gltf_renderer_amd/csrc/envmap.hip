// envmap.hip -- environment-map prerequisites of the path tracer on gfx950.
//
// EnvironmentMap::CreateEnvironmentMap (Source/EnvironmentMap.cpp:84-130) minus the raster-only GGX /
// diffuse cubes:  equirect RGB32F -> RGBA16F cube (ConvertEquirectangularToCubemap.cs.hlsl) -> cube mip
// chain (GenerateMipLevelArray.cs.hlsl) -> 1024^2 luminance importance map
// (GenerateEnvironmentImportanceMap.cs.hlsl) -> sum pyramid (GenerateEnvironmentImportanceMapLevel.cs.hlsl).
// One-off streaming kernels, one lane per output texel, coalesced along x.
#include "pt_shading.h"
#include "pt_host.h"

namespace pt {

__device__ __forceinline__ vec3 equirect_bilinear(const float* __restrict__ img, int w, int h, float u, float v) {
    float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;             // static sampler s1: linear, wrap
    float fx0 = floorf(x), fy0 = floorf(y);
    float fx = x - fx0, fy = y - fy0;
    int i0 = wrap_addr((int)fx0, w, PT_ADDRESS_WRAP), i1 = wrap_addr((int)fx0 + 1, w, PT_ADDRESS_WRAP);
    int j0 = wrap_addr((int)fy0, h, PT_ADDRESS_WRAP), j1 = wrap_addr((int)fy0 + 1, h, PT_ADDRESS_WRAP);
    float w00 = (1 - fx) * (1 - fy), w10 = fx * (1 - fy), w01 = (1 - fx) * fy, w11 = fx * fy;
    const float *a = img + ((size_t)j0 * w + i0) * 3, *b = img + ((size_t)j0 * w + i1) * 3, *c = img + ((size_t)j1 * w + i0) * 3,
                *d = img + ((size_t)j1 * w + i1) * 3;
    return v3p(a) * w00 + v3p(b) * w10 + v3p(c) * w01 + v3p(d) * w11;
}

// ConvertEquirectangularToCubemap.cs.hlsl:11-72
__global__ __launch_bounds__(256) void k_equirect_to_cube(const float* __restrict__ img, int w, int h, uint16_t* __restrict__ cube, int n) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, face = blockIdx.z;
    if (x >= n) return;
    float u = ((float)x + .5f) / (float)n, v = ((float)y + .5f) / (float)n;                   // PixelToUV
    vec3 d = cubemap_to_direction(face, u, v);
    float eu = pt_atan2(d.y, d.x) / 6.28318530717f, ev = 1 - ((d.z + 1) / 2);                   // equal-area in z (quirk q8)
    vec3 c = equirect_bilinear(img, w, h, eu, ev);
    __half hx = __float2half_rn(c.x), hy = __float2half_rn(c.y), hz = __float2half_rn(c.z), hw = __float2half_rn(1.0f);
    uint2 q = make_uint2((uint32_t)__half_as_ushort(hx) | ((uint32_t)__half_as_ushort(hy) << 16),
                         (uint32_t)__half_as_ushort(hz) | ((uint32_t)__half_as_ushort(hw) << 16));
    *(uint2*)(cube + (((size_t)face * n + y) * n + x) * 4) = q;
}

// GenerateMipLevelArray.cs.hlsl:8-30 (RWTexture2DArray<float3>: alpha is written as 0)
__global__ __launch_bounds__(256) void k_cube_mip(const uint16_t* __restrict__ in, int pn, uint16_t* __restrict__ out, int n) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, face = blockIdx.z;
    if (x >= n) return;
    vec3 r = v3(0);
    r += cube_texel(in, pn, face, 2 * x, 2 * y);
    r += cube_texel(in, pn, face, 2 * x + 1, 2 * y);
    r += cube_texel(in, pn, face, 2 * x, 2 * y + 1);
    r += cube_texel(in, pn, face, 2 * x + 1, 2 * y + 1);
    r *= 0.25f;
    uint2 q = make_uint2((uint32_t)__half_as_ushort(__float2half_rn(r.x)) | ((uint32_t)__half_as_ushort(__float2half_rn(r.y)) << 16),
                         (uint32_t)__half_as_ushort(__float2half_rn(r.z)));
    *(uint2*)(out + (((size_t)face * n + y) * n + x) * 4) = q;
}

// GenerateEnvironmentImportanceMap.cs.hlsl:13-39: trilinear cube lookup along SquareToSphere(texel)
__global__ __launch_bounds__(256) void k_importance0(const uint16_t* __restrict__ cube0, int n0, const uint16_t* __restrict__ cube1, int n1, float frac,
                                                     float* __restrict__ out, int res) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= res) return;
    vec2 uv = {((float)x + .5f) / (float)res, ((float)y + .5f) / (float)res};
    vec3 d = square_to_sphere(uv_to_square(uv));
    vec3 c = sample_cube(cube0, n0, d);
    if (frac != 0.0f) {
        vec3 c1 = sample_cube(cube1, n1, d);
        c = c * (1 - frac) + c1 * frac;
    }
    out[(size_t)y * res + x] = luminance(c);
}

// GenerateEnvironmentImportanceMapLevel.cs.hlsl:12-31: 2x2 SUM pyramid
__global__ __launch_bounds__(256) void k_importance_level(const float* __restrict__ in, float* __restrict__ out, int n) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= n) return;
    int pn = n * 2;
    float sum = in[(size_t)(2 * y) * pn + 2 * x];
    sum += in[(size_t)(2 * y + 1) * pn + 2 * x];
    sum += in[(size_t)(2 * y) * pn + 2 * x + 1];
    sum += in[(size_t)(2 * y + 1) * pn + 2 * x + 1];
    out[(size_t)y * n + x] = sum;
}

// 4x4-blocked copy of one pyramid level: texel (x, y) -> block (x/4, y/4), position (y%4)*4 + x%4 inside it.
__global__ __launch_bounds__(256) void k_importance_block(const float* __restrict__ in, float* __restrict__ out, int n) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= n) return;
    out[((size_t)(y >> 2) * (n >> 2) + (x >> 2)) * 16 + (y & 3) * 4 + (x & 3)] = in[(size_t)y * n + x];
}

hipError_t env_build(EnvDevice& e, const float* d_equirect, int w, int h, hipStream_t stream) {
    int N = (w / 4) / 2;
    N = (N > 1 ? N : 1) + 1;                                     // EnvironmentMap.cpp:92 (quirk q11)
    e.mips = 0;
    size_t off = 0;
    for (int n = N;; n >>= 1) {
        e.mip_n[e.mips] = n; e.mip_offset[e.mips] = off; off += (size_t)6 * n * n * 4; e.mips++;
        if (n <= 1) break;
    }
    hipError_t err;
    if ((err = hipMalloc(&e.cube, off * 2))) return err;
    e.levels = 0;
    uint32_t lo = 0;
    for (int r = e.imp_res; r >= 1; r >>= 1) { e.level_offset[e.levels++] = lo; lo += (uint32_t)r * r; }
    if ((err = hipMalloc(&e.importance, (size_t)lo * 4))) return err;
    hipLaunchKernelGGL(k_equirect_to_cube, dim3((N + 255) / 256, N, 6), dim3(256), 0, stream, d_equirect, w, h, e.cube, N);
    for (int l = 1; l < e.mips; l++) {
        int n = e.mip_n[l];
        hipLaunchKernelGGL(k_cube_mip, dim3((n + 255) / 256, n, 6), dim3(256), 0, stream, e.cube + e.mip_offset[l - 1], e.mip_n[l - 1],
                           e.cube + e.mip_offset[l], n);
    }
    // mip_level = clamp(log2((6*N)/1024), 0, mips) with INTEGER division (quirk q10)
    float level = log2f((float)((6u * (uint32_t)N) / (uint32_t)e.imp_res));
    level = level < 0 || !(level == level) ? 0.f : level;       // log2(0) = -inf clamps to 0
    if (level > (float)(e.mips - 1)) level = (float)(e.mips - 1);
    int l0 = (int)floorf(level), l1 = l0 + 1 < e.mips ? l0 + 1 : l0;
    float frac = l1 == l0 ? 0.f : level - (float)l0;
    hipLaunchKernelGGL(k_importance0, dim3((e.imp_res + 255) / 256, e.imp_res), dim3(256), 0, stream, e.cube + e.mip_offset[l0], e.mip_n[l0],
                       e.cube + e.mip_offset[l1], e.mip_n[l1], frac, e.importance, e.imp_res);
    for (int l = 1; l < e.levels; l++) {
        int n = e.imp_res >> l;
        hipLaunchKernelGGL(k_importance_level, dim3((n + 255) / 256, n), dim3(256), 0, stream, e.importance + e.level_offset[l - 1],
                           e.importance + e.level_offset[l], n);
    }
    // blocked copies for the two-levels-per-fetch sampling descent (pt_shading.h sample_importance_map)
    static_assert(sizeof(e.blocked_offset) / sizeof(e.blocked_offset[0]) == 5, "five level pairs");
    if (e.imp_res != 1024 || e.levels != 11) return hipErrorInvalidValue;       // IMPORTANCE_MAP_SIZE is a constant of the reference
    uint32_t bo = 0;
    for (int k = 0; k < 5; k++) { int n = 4 << (2 * k); e.blocked_offset[k] = bo; bo += (uint32_t)n * n; }
    if ((err = hipMalloc(&e.blocked, (size_t)bo * 4))) return err;
    for (int k = 0; k < 5; k++) {
        int n = 4 << (2 * k), l = 8 - 2 * k;                     // level l has resolution 1024 >> l
        hipLaunchKernelGGL(k_importance_block, dim3((n + 255) / 256, n), dim3(256), 0, stream, e.importance + e.level_offset[l],
                           e.blocked + e.blocked_offset[k], n);
    }
    if ((err = hipMemcpyAsync(&e.total, e.importance + e.level_offset[e.levels - 1], 4, hipMemcpyDeviceToHost, stream))) return err;
    if ((err = hipStreamSynchronize(stream))) return err;
    return hipGetLastError();
}

void env_free(EnvDevice& e) {
    hipFree(e.cube); hipFree(e.importance); hipFree(e.blocked);
    e.cube = nullptr; e.importance = nullptr; e.blocked = nullptr;
}

}  // namespace pt
