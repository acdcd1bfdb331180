// quadload_probe.hip -- microbenchmark behind the traversal stage's node fetch.
// Question: a lane that fetches its own 64-B node issues four dwordx4 loads, i.e. four address lookups per node; if the four lanes of
// a quad fetch ONE node together (lane k the k-th 16-B piece: one contiguous 64-B access per quad and instruction), is the wave's
// instruction cheaper for the texture addresser / L1?  Random dependent walks over a table of 64-B nodes, 6 waves per SIMD.
//   mode 0: own node, 4 x dwordx4 per lane          mode 1: quad-cooperative, 4 rounds + 4x4 quad transpose with DPP selects
//   mode 2: quad-cooperative without the transpose (load cost alone; walks with whatever piece the lane holds)
// build: hipcc -O3 --offload-arch=gfx950 -o quadload_probe tools/probes/quadload_probe.hip      run: ./quadload_probe [MB] [steps] [fillers]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int CTRL> __device__ __forceinline__ uint32_t dpp(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true); }
constexpr int QP(int a, int b, int c, int d) { return a | (b << 2) | (c << 4) | (d << 6); }

template <int FILL> __device__ __forceinline__ float filler(float x) {
#pragma unroll
    for (int i = 0; i < FILL; i++) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x));
    return x;
}

// a node: 16 dwords; dword 4..7 = four "child" indices (random), the rest payload
template <int MODE, int FILL>
__global__ __launch_bounds__(256) void k_walk(const uint4* __restrict__ nodes, uint32_t n_nodes, int steps, uint32_t* __restrict__ out, int active_mod) {
    extern __shared__ int lds_pad[];
    const uint32_t lane = threadIdx.x & 63, k = lane & 3;
    uint32_t cur = (blockIdx.x * 256u + threadIdx.x) * 2654435761u % n_nodes;
    uint32_t acc = 0;
    const bool odd = k & 1, hi = k & 2;
    // a fraction of the lanes sits out (the traversal's node phase runs with ~46 of 64 lanes): lanes with (lane % 8) >= active_mod idle
    const bool live = (lane & 7) < (uint32_t)active_mod;
    for (int s = 0; s < steps; s++) {
        uint4 p0, p1, p2, p3;
        if (MODE == 0) {
            if (live) {
                const uint4* np = nodes + (size_t)cur * 4;
                p0 = np[0]; p1 = np[1]; p2 = np[2]; p3 = np[3];
            } else { p0 = p1 = p2 = p3 = make_uint4(0, 0, 0, 0); }
        } else {
            const uint32_t mine = live ? cur : 0xffffffffu;
            uint4 r0 = make_uint4(0, 0, 0, 0), r1 = r0, r2 = r0, r3 = r0;
            const uint32_t i0 = dpp<QP(0, 0, 0, 0)>(mine), i1 = dpp<QP(1, 1, 1, 1)>(mine), i2 = dpp<QP(2, 2, 2, 2)>(mine), i3 = dpp<QP(3, 3, 3, 3)>(mine);
            if (i0 != 0xffffffffu) r0 = nodes[(size_t)i0 * 4 + k];
            if (i1 != 0xffffffffu) r1 = nodes[(size_t)i1 * 4 + k];
            if (i2 != 0xffffffffu) r2 = nodes[(size_t)i2 * 4 + k];
            if (i3 != 0xffffffffu) r3 = nodes[(size_t)i3 * 4 + k];
            if (MODE == 1) {
                // 4x4 transpose inside the quad: lane k ends up with pieces 0..3 of ITS node (register m of lane p -> register p of lane m)
#define XSTAGE(A, B, C, D, CTRL, SEL)                                               \
                { const uint32_t da_ = dpp<CTRL>(A), db_ = dpp<CTRL>(B), dc_ = dpp<CTRL>(C), dd_ = dpp<CTRL>(D);  /* every lane executes the exchange */ \
                  A = SEL ? db_ : A; B = SEL ? B : da_; C = SEL ? dd_ : C; D = SEL ? D : dc_; }
                // stage 1 (lane bit 0): register pairs (0,1), (2,3)
                XSTAGE(r0.x, r1.x, r2.x, r3.x, QP(1, 0, 3, 2), odd) XSTAGE(r0.y, r1.y, r2.y, r3.y, QP(1, 0, 3, 2), odd)
                XSTAGE(r0.z, r1.z, r2.z, r3.z, QP(1, 0, 3, 2), odd) XSTAGE(r0.w, r1.w, r2.w, r3.w, QP(1, 0, 3, 2), odd)
                // stage 2 (lane bit 1): register pairs (0,2), (1,3)
                XSTAGE(r0.x, r2.x, r1.x, r3.x, QP(2, 3, 0, 1), hi) XSTAGE(r0.y, r2.y, r1.y, r3.y, QP(2, 3, 0, 1), hi)
                XSTAGE(r0.z, r2.z, r1.z, r3.z, QP(2, 3, 0, 1), hi) XSTAGE(r0.w, r2.w, r1.w, r3.w, QP(2, 3, 0, 1), hi)
#undef XSTAGE
            }
            p0 = r0; p1 = r1; p2 = r2; p3 = r3;
        }
        // "slab tests": FILL dependent VALU instructions on the payload, then the next node from one of the four child words
        float f = filler<FILL>(__uint_as_float((p0.x ^ p2.y ^ p3.z) & 0x3fffffffu));
        acc += __float_as_uint(f) + p0.y + p2.x + p3.w;
        const uint32_t pick = (acc >> 7) & 3u;
        uint32_t nxt = pick == 0 ? p1.x : (pick == 1 ? p1.y : (pick == 2 ? p1.z : p1.w));
        nxt ^= acc * 2654435761u;                                    // uniform pseudo-random successor in every mode (mode 2's pieces are not the lane's own)
        if (live) cur = nxt % n_nodes;
    }
    if (acc == 0x12345678u) out[0] = acc;
    if (threadIdx.x == 0 && lds_pad[0] == 77) out[1] = 1;
}

// the transpose against the plain loads: mismatching dwords
__global__ __launch_bounds__(256) void k_check(const uint4* __restrict__ nodes, uint32_t n_nodes, uint32_t* __restrict__ bad) {
    const uint32_t lane = threadIdx.x & 63, k = lane & 3;
    const bool odd = k & 1, hi = k & 2;
    const uint32_t cur = (blockIdx.x * 256u + threadIdx.x) * 2654435761u % n_nodes;
    const uint4* np = nodes + (size_t)cur * 4;
    const uint4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
    const uint32_t i0 = dpp<QP(0, 0, 0, 0)>(cur), i1 = dpp<QP(1, 1, 1, 1)>(cur), i2 = dpp<QP(2, 2, 2, 2)>(cur), i3 = dpp<QP(3, 3, 3, 3)>(cur);
    uint4 r0 = nodes[(size_t)i0 * 4 + k], r1 = nodes[(size_t)i1 * 4 + k], r2 = nodes[(size_t)i2 * 4 + k], r3 = nodes[(size_t)i3 * 4 + k];
#define XSTAGE(A, B, C, D, CTRL, SEL)                                               \
    { const uint32_t da_ = dpp<CTRL>(A), db_ = dpp<CTRL>(B), dc_ = dpp<CTRL>(C), dd_ = dpp<CTRL>(D);      \
      A = SEL ? db_ : A; B = SEL ? B : da_; C = SEL ? dd_ : C; D = SEL ? D : dc_; }
    XSTAGE(r0.x, r1.x, r2.x, r3.x, QP(1, 0, 3, 2), odd) XSTAGE(r0.y, r1.y, r2.y, r3.y, QP(1, 0, 3, 2), odd)
    XSTAGE(r0.z, r1.z, r2.z, r3.z, QP(1, 0, 3, 2), odd) XSTAGE(r0.w, r1.w, r2.w, r3.w, QP(1, 0, 3, 2), odd)
    XSTAGE(r0.x, r2.x, r1.x, r3.x, QP(2, 3, 0, 1), hi) XSTAGE(r0.y, r2.y, r1.y, r3.y, QP(2, 3, 0, 1), hi)
    XSTAGE(r0.z, r2.z, r1.z, r3.z, QP(2, 3, 0, 1), hi) XSTAGE(r0.w, r2.w, r1.w, r3.w, QP(2, 3, 0, 1), hi)
#undef XSTAGE
    auto ne = [](uint4 a, uint4 b) { return (a.x != b.x) + (a.y != b.y) + (a.z != b.z) + (a.w != b.w); };
    const int m = ne(q0, r0) + ne(q1, r1) + ne(q2, r2) + ne(q3, r3);
    if (m) atomicAdd(bad, (uint32_t)m);
}

static uint32_t rng(uint32_t& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

template <int MODE, int FILL>
static void run(const char* name, const uint4* d_nodes, uint32_t n, int steps, uint32_t* d_out, int active_mod) {
    const int blocks = 256 * 6;           // 6 workgroups of 4 waves per CU = 6 waves per SIMD (LDS pad caps it there)
    const size_t lds = 24 * 1024;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_walk<MODE, FILL>), dim3(blocks), dim3(256), lds, 0, d_nodes, n, steps / 4, d_out, active_mod);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_walk<MODE, FILL>), dim3(blocks), dim3(256), lds, 0, d_nodes, n, steps, d_out, active_mod);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    const double lane_steps = (double)blocks * 256 * steps * active_mod / 8.0;
    printf("%-44s fill %3d active %d/8: %8.3f ms  %7.1f G node fetches/s  (%.1f TB/s of node bytes)\n", name, FILL, active_mod, ms, lane_steps / ms / 1e6,
           lane_steps * 64 / ms / 1e9);
}

int main(int argc, char** argv) {
    const size_t mb = argc > 1 ? atoi(argv[1]) : 32;
    const int steps = argc > 2 ? atoi(argv[2]) : 400;
    const uint32_t n = (uint32_t)(mb * 1024 * 1024 / 64);
    std::vector<uint32_t> h((size_t)n * 16);
    uint32_t s = 12345;
    for (auto& v : h) v = rng(s);
    uint4* d_nodes; uint32_t* d_out;
    CK(hipMalloc(&d_nodes, (size_t)n * 64)); CK(hipMalloc(&d_out, 64));
    CK(hipMemcpy(d_nodes, h.data(), (size_t)n * 64, hipMemcpyHostToDevice));
    printf("table %zu MB (%u nodes), %d steps per lane\n", mb, n, steps);
    CK(hipMemset(d_out, 0, 64));
    hipLaunchKernelGGL(k_check, dim3(1024), dim3(256), 0, 0, d_nodes, n, d_out + 4);
    uint32_t bad = 0; CK(hipMemcpy(&bad, d_out + 4, 4, hipMemcpyDeviceToHost));
    printf("transpose check: %u mismatching dwords\n", bad);
    for (int am : {8, 6}) {
        run<0, 0>("own node, 4 x dwordx4 per lane", d_nodes, n, steps, d_out, am);
        run<2, 0>("quad-cooperative, no transpose", d_nodes, n, steps, d_out, am);
        run<1, 0>("quad-cooperative + DPP transpose", d_nodes, n, steps, d_out, am);
        run<0, 120>("own node, 4 x dwordx4 per lane", d_nodes, n, steps, d_out, am);
        run<2, 120>("quad-cooperative, no transpose", d_nodes, n, steps, d_out, am);
        run<1, 120>("quad-cooperative + DPP transpose", d_nodes, n, steps, d_out, am);
    }
    return 0;
}
