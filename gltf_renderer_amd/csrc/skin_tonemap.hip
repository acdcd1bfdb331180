// skin_tonemap.hip -- GpuSkin and ToneMapper on gfx950.
//
// k_skin:     Source/Shaders/Skin.cs.hlsl:53-136 as dispatched by GpuSkin::Run (Source/GpuSkin.cpp:57-118):
//             <=4 morph targets, 4-joint linear-blend skinning of position / normal / tangent, 10-10-10-2
//             tangent-space re-encode.  One lane per vertex, 64-lane groups like the reference, streams
//             read once (12+4+16 B in, 12+4 B out per vertex: HBM-bound).
// k_skin_mfma: the same blend with the joint-matrix contraction on the matrix cores: per 16-vertex tile
//             M[16 x 12] = W[16 x J] * B[J x 12] via v_mfma_f32_16x16x4_f32 (exact f32), W densified from
//             the 4 (joint, weight) pairs per vertex; then position/normal/tangent are transformed by the
//             blended 3x4 matrices.  Blending matrices first and transforming once differs from the
//             reference's "transform four times, then blend" only by fp32 re-association (~1 ulp).
// k_tonemap:  Source/Shaders/ToneMapper.ps.hlsl:30-101 (exposure, AgX or clamp, sRGB OETF, optional dither).
#include "pt_shading.h"
#include "pt_host.h"

namespace pt {


__device__ __forceinline__ void skin_load_morph(const SkinArgs& a, uint32_t index, vec3& position, vec3& normal, vec3& tangent, float& winding) {
    position = v3p(a.in_position + (size_t)index * 3);
    normal = v3(0); tangent = v3(0); winding = 1;
    if (a.input_mesh_flags & PT_MESH_FLAG_TANGENT_SPACE) decode_tangent_space(a.in_tangent_space[index], normal, tangent, winding);
    for (int i = 0; i < a.num_of_morph_targets; i++) {                      // Skin.cs.hlsl:71-88
        float w = a.morph_weight[i];
        if (a.morph_position[i]) position += w * v3p(a.morph_position[i] + (size_t)index * 3);
        if (a.morph_tangent_space[i]) {
            vec3 mn, mt; float mw;
            decode_tangent_space(a.morph_tangent_space[i][index], mn, mt, mw);
            normal += w * mn;
            tangent += w * mt;
        }
    }
}
__device__ __forceinline__ void skin_store(const SkinArgs& a, uint32_t index, vec3 position, vec3 normal, vec3 tangent, float winding) {
    if (a.output_mesh_flags & PT_DYNAMIC_MESH_FLAG_POSITION) {
        float* o = a.out_position + (size_t)index * 3;
        o[0] = position.x; o[1] = position.y; o[2] = position.z;
    }
    if (a.output_mesh_flags & PT_DYNAMIC_MESH_FLAG_TANGENT_SPACE)
        a.out_tangent_space[index] = encode_tangent_space(normalize(normal), normalize(tangent), winding);
}

__global__ __launch_bounds__(64) void k_skin(SkinArgs a) {
    uint32_t index = blockIdx.x * 64 + threadIdx.x;
    if (index >= a.num_of_vertices) return;
    vec3 position, normal, tangent; float winding;
    skin_load_morph(a, index, position, normal, tangent, winding);
    if (a.input_mesh_flags & PT_MESH_FLAG_JOINT_WEIGHT) {                   // :91-128
        uint4 bw = a.in_joint_weight[index];
        uint32_t ids[4] = {bw.x & 0xffff, bw.x >> 16, bw.y & 0xffff, bw.y >> 16};
        float w[4] = {(float)(bw.z & 0xffff) / 65535.0f, (float)(bw.z >> 16) / 65535.0f, (float)(bw.w & 0xffff) / 65535.0f, (float)(bw.w >> 16) / 65535.0f};
        // A joint id beyond the bone array reads zeros upstream (D3D12 robust buffer access on the structured buffer,
        // Skin.cs.hlsl:93-101): a zero matrix, i.e. no contribution.  Here it must not become a raw out-of-bounds load.
        const uint32_t nb = (uint32_t)a.bone_count;
        vec3 sp = v3(0);
#pragma unroll
        for (int i = 0; i < 4; i++) if (ids[i] < nb) sp += w[i] * mul_point(a.bones[ids[i]].transform, position);
        position = sp;
        if (a.input_mesh_flags & PT_MESH_FLAG_TANGENT_SPACE) {
            vec3 sn = v3(0), stn = v3(0);
#pragma unroll
            for (int i = 0; i < 4; i++) if (ids[i] < nb) sn += w[i] * mul_dir(a.bones[ids[i]].inverse_transpose, normal);
#pragma unroll
            for (int i = 0; i < 4; i++) if (ids[i] < nb) stn += w[i] * mul_dir(a.bones[ids[i]].transform, tangent);
            normal = sn; tangent = stn;
        }
    }
    skin_store(a, index, position, normal, tangent, winding);
}

// ---- MFMA joint-matrix blend ----------------------------------------------------------------------------
// One wave per 16 vertices.  v_mfma_f32_16x16x4_f32: D[16x16] += A[16x4] * B[4x16]; lane l holds
// A[l%16][l/16] and B[l/16][l%16]; D: lane l holds D[4*(l/16) + r][l%16] in register r.
// A = densified weights (row = vertex, col = joint of the current K-slab), B = bone matrix entries
// (row = joint, col = one of 12 affine entries; the 3x3 of inverse_transpose go in a second pass).
typedef float floatx4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(64) void k_skin_mfma(SkinArgs a) {
    __shared__ float s_m[2][16][16];                       // blended matrices: [0] transform (12 used), [1] inverse_transpose (9 used)
    const uint32_t lane = threadIdx.x, row = lane & 15, kq = lane >> 4;
    const uint32_t base = blockIdx.x * 16;
    const uint32_t vtx = base + row;
    uint32_t ids[4] = {0, 0, 0, 0};
    float w[4] = {0, 0, 0, 0};
    if (vtx < a.num_of_vertices) {
        uint4 bw = a.in_joint_weight[vtx];
        ids[0] = bw.x & 0xffff; ids[1] = bw.x >> 16; ids[2] = bw.y & 0xffff; ids[3] = bw.y >> 16;
        w[0] = (float)(bw.z & 0xffff) / 65535.0f; w[1] = (float)(bw.z >> 16) / 65535.0f;
        w[2] = (float)(bw.w & 0xffff) / 65535.0f; w[3] = (float)(bw.w >> 16) / 65535.0f;
    }
    floatx4 acc_t = {0, 0, 0, 0}, acc_n = {0, 0, 0, 0};
    const int slabs = (a.bone_count + 3) / 4;
    for (int s = 0; s < slabs; s++) {
        const uint32_t joint = (uint32_t)s * 4 + kq;       // column of A / row of B held by this lane
        float av = 0.f;                                    // W[row][joint]: duplicates of a joint add up, as in the reference's sum
#pragma unroll
        for (int i = 0; i < 4; i++) av += (ids[i] == joint) ? w[i] : 0.f;
        float bt = 0.f, bn = 0.f;
        if (joint < (uint32_t)a.bone_count) {
            // column c (= row of this lane's B element): affine entry c -> transform[(c/3)*4 + c%3] for c<12
            const uint32_t c = row;
            if (c < 12) bt = a.bones[joint].transform[(c / 3) * 4 + (c % 3)];
            if (c < 9) bn = a.bones[joint].inverse_transpose[(c / 3) * 4 + (c % 3)];
        }
        acc_t = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bt, acc_t, 0, 0, 0);
        acc_n = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bn, acc_n, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) { s_m[0][kq * 4 + r][row] = acc_t[r]; s_m[1][kq * 4 + r][row] = acc_n[r]; }
    __syncthreads();
    if (lane < 16 && base + lane < a.num_of_vertices) {
        const uint32_t index = base + lane;
        vec3 position, normal, tangent; float winding;
        skin_load_morph(a, index, position, normal, tangent, winding);
        const float* T = s_m[0][lane];                     // 12 entries: column-major 3x4 (col*3 + row)
        const float* Nm = s_m[1][lane];
        vec3 p = v3(T[0] * position.x + T[3] * position.y + T[6] * position.z + T[9],
                    T[1] * position.x + T[4] * position.y + T[7] * position.z + T[10],
                    T[2] * position.x + T[5] * position.y + T[8] * position.z + T[11]);
        if (a.input_mesh_flags & PT_MESH_FLAG_TANGENT_SPACE) {
            vec3 n2 = v3(Nm[0] * normal.x + Nm[3] * normal.y + Nm[6] * normal.z, Nm[1] * normal.x + Nm[4] * normal.y + Nm[7] * normal.z,
                         Nm[2] * normal.x + Nm[5] * normal.y + Nm[8] * normal.z);
            vec3 t2 = v3(T[0] * tangent.x + T[3] * tangent.y + T[6] * tangent.z, T[1] * tangent.x + T[4] * tangent.y + T[7] * tangent.z,
                         T[2] * tangent.x + T[5] * tangent.y + T[8] * tangent.z);
            normal = n2; tangent = t2;
        }
        skin_store(a, index, p, normal, tangent, winding);
    }
}

void launch_skin(const SkinArgs& a, bool use_mfma, hipStream_t stream) {
    if (a.num_of_vertices == 0) return;
    bool skinned = (a.input_mesh_flags & PT_MESH_FLAG_JOINT_WEIGHT) != 0;
    if (use_mfma && skinned) hipLaunchKernelGGL(k_skin_mfma, dim3((a.num_of_vertices + 15) / 16), dim3(64), 0, stream, a);
    else hipLaunchKernelGGL(k_skin, dim3((a.num_of_vertices + 63) / 64), dim3(64), 0, stream, a);   // GpuSkin.cpp:108
}

// ---- tone mapper --------------------------------------------------------------------------------------------
__device__ __forceinline__ vec3 agx_curve(vec3 x) {                        // ToneMapper.ps.hlsl:30-44
    vec3 x2 = x * x, x4 = x2 * x2;
    vec3 r = 15.5f * x4 * x2;
    r = r - 40.14f * x4 * x;
    r = r + 31.96f * x4;
    r = r - 6.868f * x2 * x;
    r = r + 0.4298f * x2;
    r = r + 0.1191f * x;
    r = r - 0.00232f;
    return r;
}
__device__ __forceinline__ vec3 agx_tonemap(vec3 c) {                      // :49-75
    vec3 i = v3(0.856627153315983f * c.x + 0.0951212405381588f * c.y + 0.0482516061458583f * c.z,
                0.137318972929847f * c.x + 0.761241990602591f * c.y + 0.101439036467562f * c.z,
                0.11189821299995f * c.x + 0.0767994186031903f * c.y + 0.811302368396859f * c.z);
    const float log_min = -12.47393f, log_max = 4.026069f;
    i = v3(clampf(co_log2(i.x), log_min, log_max), clampf(co_log2(i.y), log_min, log_max), clampf(co_log2(i.z), log_min, log_max));
    i = (i - log_min) / (log_max - log_min);
    i = agx_curve(i);
    vec3 o = v3(1.12710058f * i.x + -0.11060664f * i.y + -0.01649394f * i.z, -0.14132976f * i.x + 1.1578237f * i.y + -0.01649394f * i.z,
                -0.14132976f * i.x + -0.11060664f * i.y + 1.25193641f * i.z);
    return v3(hpow(o.x, 2.2f), hpow(o.y, 2.2f), hpow(o.z, 2.2f));
}
__device__ __forceinline__ float srgb_oetf(float x) { return x <= 0.0031308f ? x * 12.92f : 1.055f * hpow(x, 1.f / 2.4f) - 0.055f; }   // Color.hlsli:9-17
__device__ __forceinline__ void pcg3d(uint32_t& x, uint32_t& y, uint32_t& z) {   // Random.hlsli:3-15
    x = x * 1664525u + 1013904223u; y = y * 1664525u + 1013904223u; z = z * 1664525u + 1013904223u;
    x += y * z; y += z * x; z += x * y;
    x ^= x >> 16; y ^= y >> 16; z ^= z >> 16;
    x += y * z; y += z * x; z += x * y;
}

__global__ __launch_bounds__(256) void k_tonemap(const float4* __restrict__ in, uint32_t w, uint32_t h, pt_tonemap_config cfg, float* __restrict__ out_rgb,
                                                 uint32_t* __restrict__ out_rgba8) {
    uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    // :87-88  pixel = UVToPixel(uv, resolution): (int)(floor(uv*res) - 0.5), one texel off for k >= 1 (quirk q9)
    float u = ((float)x + 0.5f) / (float)w, v = ((float)y + 0.5f) / (float)h;
    int sx = f2i(floorf(u * (float)w) - .5f), sy = f2i(floorf(v * (float)h) - .5f);
    float4 s = in[(size_t)sy * w + sx];
    vec3 c = cfg.exposure * v3(s.x, s.y, s.z);
    if (cfg.tonemapper == PT_TONEMAPPER_NONE) c = v3(saturate(c.x), saturate(c.y), saturate(c.z));
    else c = agx_tonemap(c);
    c = v3(srgb_oetf(c.x), srgb_oetf(c.y), srgb_oetf(c.z));
    if (cfg.dither) {                                                      // :77-81
        uint32_t a0 = (uint32_t)sx * 2, a1 = (uint32_t)sy * 2, a2 = (uint32_t)cfg.frame * 2, b0 = a0 + 1, b1 = a1 + 1, b2 = a2 + 1;
        pcg3d(a0, a1, a2); pcg3d(b0, b1, b2);
        const float sc = 2.3283064365386963e-10f;
        c = c + v3((float)a0 * sc + (float)b0 * sc - 1.0f, (float)a1 * sc + (float)b1 * sc - 1.0f, (float)a2 * sc + (float)b2 * sc - 1.0f) / 255.f;
    }
    size_t i = (size_t)y * w + x;
    if (out_rgb) { out_rgb[i * 3] = c.x; out_rgb[i * 3 + 1] = c.y; out_rgb[i * 3 + 2] = c.z; }
    if (out_rgba8) {
        uint32_t r = (uint32_t)(saturate(c.x) * 255.f + 0.5f), g = (uint32_t)(saturate(c.y) * 255.f + 0.5f), b = (uint32_t)(saturate(c.z) * 255.f + 0.5f);
        out_rgba8[i] = r | (g << 8) | (b << 16) | 0xff000000u;
    }
}

void launch_tonemap(const float4* in, uint32_t w, uint32_t h, const pt_tonemap_config& cfg, float* out_rgb, uint32_t* out_rgba8, hipStream_t stream) {
    hipLaunchKernelGGL(k_tonemap, dim3((w + 255) / 256, h), dim3(256), 0, stream, in, w, h, cfg, out_rgb, out_rgba8);
}

}  // namespace pt
