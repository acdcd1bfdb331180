// random_sector_probe.hip -- what the memory system sustains for the shade stage's access pattern: every lane
// gathers 8-B / 16-B pieces at random 64-B-aligned addresses of a buffer far larger than the Infinity Cache (so every piece costs a
// whole 64-B sector from HBM), B independent gathers in flight per lane and round, at the shade stage's occupancy (2 workgroups of
// 4 waves per CU) and at the traversal stages' (6).
// build: hipcc -O3 --offload-arch=gfx950 -o random_sector_probe tools/probes/random_sector_probe.hip     run: ./random_sector_probe [GB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t pcg(uint32_t& s) { s = s * 747796405u + 2891336453u; uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u; return (w >> 22u) ^ w; }

// BYTES == 32: two 16-B pieces of ONE random 128-B-aligned block (offsets 0 and 64): does the second half of a line cost anything?
template <int B, int BYTES>
__global__ __launch_bounds__(256) void k_gather(const char* __restrict__ buf, uint64_t sectors, int rounds, uint32_t* __restrict__ out) {
    extern __shared__ int lds_pad[];
    uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    for (int r = 0; r < rounds; r++) {
        uint64_t a[B];
#pragma unroll
        for (int k = 0; k < B; k++) { const uint64_t x = ((uint64_t)pcg(s) << 20) ^ pcg(s); a[k] = BYTES == 32 ? (x % (sectors / 2)) * 128u : (x % sectors) * 64u; }
        if (BYTES == 32) {
            uint4 v[B], w[B];
#pragma unroll
            for (int k = 0; k < B; k++) { v[k] = *(const uint4*)(buf + a[k]); w[k] = *(const uint4*)(buf + a[k] + 64); }
#pragma unroll
            for (int k = 0; k < B; k++) acc += v[k].x ^ v[k].y ^ w[k].z ^ w[k].w;
        } else if (BYTES == 8) {
            uint2 v[B];
#pragma unroll
            for (int k = 0; k < B; k++) v[k] = *(const uint2*)(buf + a[k]);
#pragma unroll
            for (int k = 0; k < B; k++) acc += v[k].x ^ v[k].y;
        } else {
            uint4 v[B];
#pragma unroll
            for (int k = 0; k < B; k++) v[k] = *(const uint4*)(buf + a[k]);
#pragma unroll
            for (int k = 0; k < B; k++) acc += v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
        }
        s ^= acc & 1u;                   // the next round's addresses depend on this round's data: dependent rounds, like a hit's fetches
    }
    if (acc == 0x12345678u) out[0] = acc;
    if (threadIdx.x == 0 && lds_pad[0] == 77) out[1] = 1;
}

template <int B, int BYTES>
static void run(const char* buf, uint64_t sectors, int wg_per_cu, uint32_t* d_out) {
    const int blocks = 256 * wg_per_cu, rounds = 200;
    const size_t lds = wg_per_cu == 2 ? 68 * 1024 : 24 * 1024;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_gather<B, BYTES>), dim3(blocks), dim3(256), lds, 0, buf, sectors, rounds / 4, d_out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_gather<B, BYTES>), dim3(blocks), dim3(256), lds, 0, buf, sectors, rounds, d_out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    const double n = (double)blocks * 256 * rounds * B;
    printf("%2d waves/CU, %d x %2d-B gathers in flight per lane: %7.3f ms  %6.1f G gathers/s  = %5.2f TB/s of 64-B sectors\n", wg_per_cu * 4, B, BYTES, ms, n / ms / 1e6,
           n * 64 / ms / 1e9);
}

int main(int argc, char** argv) {
    const size_t gb = argc > 1 ? atoi(argv[1]) : 8;
    const uint64_t bytes = gb << 30, sectors = bytes / 64;
    char* buf; uint32_t* d_out;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&d_out, 64));
    CK(hipMemset(buf, 1, bytes));
    printf("buffer %zu GB\n", gb);
    for (int w : {2, 6}) {
        run<1, 8>(buf, sectors, w, d_out); run<4, 8>(buf, sectors, w, d_out); run<8, 8>(buf, sectors, w, d_out); run<16, 8>(buf, sectors, w, d_out);
        run<4, 16>(buf, sectors, w, d_out); run<8, 16>(buf, sectors, w, d_out);
        run<4, 32>(buf, sectors, w, d_out);       // "32-B gathers" = both 64-B halves of one 128-B block per gather
    }
    return 0;
}
