// sort_scan.hip -- the two data-parallel primitives of the acceleration-structure build, hand-written for gfx950 (wave64):
//
//   radix_sort_pairs   stable LSD radix sort of (u64 key, u32 value) pairs: the Morton order of the triangles (accel.hip step 3).
//                      Seven passes of 9 bits cover the 63 bits of a Morton code (21 bits per axis); per pass a tile histogram, an
//                      exclusive scan of the digit-major histogram table, and a scatter that ranks a tile's items STABLY -- striped over
//                      the workgroup, eight rounds, a wave64 match by nine ballots per round -- stages them in LDS in sorted order and
//                      writes each digit's run out contiguously.
//   exclusive_scan     u32 exclusive prefix sum (the compaction of the PLOC builder's cluster list, and the sort's own offsets):
//                      2048-element tiles scanned in LDS, tile sums scanned recursively, offsets added back.
//
// Until round 3 both came from rocPRIM -- the one library kernel family in profiles/.  Same results to the bit (the sort is stable, as
// rocPRIM's is; tests/test_gpu_round3.py compares images bit for bit and the sorted order against numpy).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "pt_host.h"

namespace pt {

namespace {

constexpr int kSsBlock = 256;                 // threads per workgroup
constexpr int kSsItems = 8;                   // items per thread
constexpr int kSsTile = kSsBlock * kSsItems;  // 2048 items per workgroup
constexpr int kRsBits = 9;
constexpr int kRsBins = 1 << kRsBits;         // 512
constexpr int kRsPasses = 7;                  // 63 bits

// ---- exclusive scan -------------------------------------------------------------------------------------------------------------
// wave64 inclusive scan by DPP-free shuffles (six steps)
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)v, d, 64);
        if (lane >= (uint32_t)d) v += o;
    }
    return v;
}
// Exclusive scan of a workgroup's 256 per-thread values; returns this thread's offset, *total = the workgroup's sum.  `s_w`: 4 words of LDS.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* s_w, uint32_t* total) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_scan(v);
    if (lane == 63u) s_w[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < wave; w++) base += s_w[w];
    *total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    __syncthreads();                           // s_w is reused by the caller
    return base + inc - v;
}

// One tile: out[i] = exclusive prefix inside the tile (blocked: thread t owns items t*8 .. t*8+7), sums[tile] = the tile's total.
__global__ __launch_bounds__(kSsBlock) void k_scan_tiles(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t* __restrict__ sums, size_t n) {
    __shared__ uint32_t s_w[4];
    const size_t base = (size_t)blockIdx.x * kSsTile + (size_t)threadIdx.x * kSsItems;
    uint32_t v[kSsItems], sum = 0;
#pragma unroll
    for (int k = 0; k < kSsItems; k++) { v[k] = base + k < n ? in[base + k] : 0u; sum += v[k]; }
    uint32_t total;
    uint32_t run = block_exclusive_scan(sum, s_w, &total);
#pragma unroll
    for (int k = 0; k < kSsItems; k++) { if (base + k < n) out[base + k] = run; run += v[k]; }
    if (threadIdx.x == 0 && sums) sums[blockIdx.x] = total;
}
__global__ __launch_bounds__(kSsBlock) void k_scan_add(uint32_t* __restrict__ out, const uint32_t* __restrict__ tile_offsets, size_t n) {
    const uint32_t add = tile_offsets[blockIdx.x];
    const size_t base = (size_t)blockIdx.x * kSsTile + (size_t)threadIdx.x * kSsItems;
#pragma unroll
    for (int k = 0; k < kSsItems; k++) if (base + k < n) out[base + k] += add;
}

size_t tiles_of(size_t n) { return (n + kSsTile - 1) / kSsTile; }

// temp layout of a scan of n elements: sums of level 0 (tiles_of(n) words), their scan's sums, ... until one tile is left
size_t scan_temp_words(size_t n) {
    size_t words = 0;
    for (size_t m = tiles_of(n); m > 1; m = tiles_of(m)) words += m + 64;
    return words + 64;
}

hipError_t scan_rec(uint32_t* temp, const uint32_t* in, uint32_t* out, size_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const size_t m = tiles_of(n);
    if (m == 1) {
        hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(kSsBlock), 0, stream, in, out, (uint32_t*)nullptr, n);
        return hipGetLastError();
    }
    uint32_t* sums = temp;                                  // m words, scanned in place
    hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)m), dim3(kSsBlock), 0, stream, in, out, sums, n);
    hipError_t e = scan_rec(temp + m + 64, sums, sums, m, stream);
    if (e) return e;
    hipLaunchKernelGGL(k_scan_add, dim3((unsigned)m), dim3(kSsBlock), 0, stream, out, (const uint32_t*)sums, n);
    return hipGetLastError();
}

// ---- radix sort ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t digit_of(uint64_t key, int shift) { return (uint32_t)(key >> shift) & (uint32_t)(kRsBins - 1); }

// tile histogram of one digit position: hist[digit * tiles + tile]
__global__ __launch_bounds__(kSsBlock) void k_rs_hist(const uint64_t* __restrict__ keys, size_t n, int shift, uint32_t tiles, uint32_t* __restrict__ hist) {
    __shared__ uint32_t s_h[kRsBins];
    for (int b = threadIdx.x; b < kRsBins; b += kSsBlock) s_h[b] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * kSsTile;
#pragma unroll
    for (int j = 0; j < kSsItems; j++) {
        const size_t i = base + (size_t)j * kSsBlock + threadIdx.x;
        if (i < n) atomicAdd(&s_h[digit_of(keys[i], shift)], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < kRsBins; b += kSsBlock) hist[(size_t)b * tiles + blockIdx.x] = s_h[b];
}

// Scatter of one pass.  `offsets` = the exclusive scan of the histogram table: where (digit, tile)'s run starts in the output.
// Items are STRIPED over the workgroup (item j of thread t is element j * 256 + t of the tile), so tile order = (round j, wave, lane) and a
// stable rank is: items of the same digit in earlier rounds + in lower waves of this round + in lower lanes of this wave.
__global__ __launch_bounds__(kSsBlock) void k_rs_scatter(const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, size_t n, int shift, uint32_t tiles,
                                                         const uint32_t* __restrict__ offsets, uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out) {
    __shared__ uint32_t s_wave[4][kRsBins];      // this round's count per wave and digit
    __shared__ uint32_t s_run[kRsBins];          // items of each digit in the rounds so far; after the rounds: the tile's histogram
    __shared__ uint32_t s_start[kRsBins];        // exclusive scan of s_run: where a digit's run starts inside the sorted tile
    __shared__ uint32_t s_w[4];
    __shared__ uint64_t s_key[kSsTile];
    __shared__ uint32_t s_val[kSsTile];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const size_t base = (size_t)blockIdx.x * kSsTile;
    for (int b = threadIdx.x; b < kRsBins; b += kSsBlock) s_run[b] = 0;
    uint64_t key[kSsItems];
    uint32_t val[kSsItems], rank[kSsItems];
#pragma unroll
    for (int j = 0; j < kSsItems; j++) {
        const size_t i = base + (size_t)j * kSsBlock + threadIdx.x;
        key[j] = i < n ? keys_in[i] : 0ull;
        val[j] = i < n ? vals_in[i] : 0u;
    }
#pragma unroll                                  // (fully unrolled: key[] / rank[] stay in registers)
    for (int j = 0; j < kSsItems; j++) {
        const bool valid = base + (size_t)j * kSsBlock + threadIdx.x < n;
        const uint32_t d = digit_of(key[j], shift);
        for (int b = threadIdx.x; b < 4 * kRsBins; b += kSsBlock) (&s_wave[0][0])[b] = 0;
        __syncthreads();
        // the lanes of this wave that hold the same digit: nine ballots
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int k = 0; k < kRsBits; k++) {
            const bool bit = (d >> k) & 1u;
            const unsigned long long b = __ballot(valid && bit);
            peers &= bit ? b : ~b;
        }
        const uint32_t in_wave = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
        if (valid && in_wave == 0) s_wave[wave][d] = (uint32_t)__popcll(peers);       // the group's first lane books the group
        __syncthreads();
        uint32_t before = s_run[d];
        for (uint32_t w = 0; w < wave; w++) before += s_wave[w][d];
        rank[j] = before + in_wave;
        __syncthreads();
        for (int b = threadIdx.x; b < kRsBins; b += kSsBlock) s_run[b] += s_wave[0][b] + s_wave[1][b] + s_wave[2][b] + s_wave[3][b];
        __syncthreads();
    }
    // where each digit's run starts inside the sorted tile: exclusive scan of the 512 counts (two per thread)
    {
        const uint32_t a = s_run[2 * threadIdx.x], b = s_run[2 * threadIdx.x + 1];
        uint32_t total;
        const uint32_t off = block_exclusive_scan(a + b, s_w, &total);
        s_start[2 * threadIdx.x] = off; s_start[2 * threadIdx.x + 1] = off + a;
    }
    __syncthreads();
    // the tile in sorted order, in LDS
#pragma unroll
    for (int j = 0; j < kSsItems; j++) {
        if (base + (size_t)j * kSsBlock + threadIdx.x < n) {
            const uint32_t p = s_start[digit_of(key[j], shift)] + rank[j];
            s_key[p] = key[j]; s_val[p] = val[j];
        }
    }
    __syncthreads();
    // ... and out: position p of the sorted tile belongs to digit d's run, which starts at offsets[d][tile] in the output
    const uint32_t count = (uint32_t)((n - base) < (size_t)kSsTile ? (n - base) : (size_t)kSsTile);
    for (uint32_t p = threadIdx.x; p < count; p += kSsBlock) {
        const uint64_t k = s_key[p];
        const uint32_t d = digit_of(k, shift);
        const size_t dst = (size_t)offsets[(size_t)d * tiles + blockIdx.x] + (p - s_start[d]);
        keys_out[dst] = k; vals_out[dst] = s_val[p];
    }
}

}  // namespace

size_t exclusive_scan_temp_bytes(size_t n) { return scan_temp_words(n) * 4; }

hipError_t exclusive_scan_u32(void* temp, const uint32_t* in, uint32_t* out, size_t n, hipStream_t stream) {
    return scan_rec((uint32_t*)temp, in, out, n, stream);
}

// temp: the histogram table (512 x tiles words), its scan (the same again) and the scan's own temporaries
size_t radix_sort_temp_bytes(size_t n) {
    const size_t table = (size_t)kRsBins * tiles_of(n);
    return (2 * table + scan_temp_words(table) + 256) * 4;
}

// Sorts n (key, value) pairs by the low 63 bits of the key, stably.  keys_in / vals_in are used as scratch (ping-pong); the result is in
// keys_out / vals_out (seven passes: an odd number).
hipError_t radix_sort_pairs_u64_u32(void* temp, uint64_t* keys_in, uint64_t* keys_out, uint32_t* vals_in, uint32_t* vals_out, size_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const size_t tiles = tiles_of(n), table = (size_t)kRsBins * tiles;
    uint32_t* hist = (uint32_t*)temp;
    uint32_t* offs = hist + table;
    uint32_t* scan_temp = offs + table;
    uint64_t* k[2] = {keys_in, keys_out};
    uint32_t* v[2] = {vals_in, vals_out};
    for (int pass = 0; pass < kRsPasses; pass++) {
        const int shift = pass * kRsBits, src = pass & 1, dst = src ^ 1;
        hipLaunchKernelGGL(k_rs_hist, dim3((unsigned)tiles), dim3(kSsBlock), 0, stream, (const uint64_t*)k[src], n, shift, (uint32_t)tiles, hist);
        hipError_t e = scan_rec(scan_temp, hist, offs, table, stream);
        if (e) return e;
        hipLaunchKernelGGL(k_rs_scatter, dim3((unsigned)tiles), dim3(kSsBlock), 0, stream, (const uint64_t*)k[src], (const uint32_t*)v[src], n, shift, (uint32_t)tiles,
                           (const uint32_t*)offs, k[dst], v[dst]);
    }
    return hipGetLastError();
}

}  // namespace pt

// ---- test hooks (not part of include/mipt.h: tests/test_gpu_round3.py drives the primitives directly, host arrays in and out) ----------------
extern "C" int pt_debug_sort_pairs(uint64_t* keys, uint32_t* vals, size_t n) {
    if (n == 0) return 0;
    uint64_t *ka = nullptr, *kb = nullptr; uint32_t *va = nullptr, *vb = nullptr; void* temp = nullptr;
    int rc = -3;
    if (hipMalloc(&ka, n * 8) == hipSuccess && hipMalloc(&kb, n * 8) == hipSuccess && hipMalloc(&va, n * 4) == hipSuccess && hipMalloc(&vb, n * 4) == hipSuccess &&
        hipMalloc(&temp, pt::radix_sort_temp_bytes(n)) == hipSuccess && hipMemcpy(ka, keys, n * 8, hipMemcpyHostToDevice) == hipSuccess &&
        hipMemcpy(va, vals, n * 4, hipMemcpyHostToDevice) == hipSuccess && pt::radix_sort_pairs_u64_u32(temp, ka, kb, va, vb, n, nullptr) == hipSuccess &&
        hipDeviceSynchronize() == hipSuccess && hipMemcpy(keys, kb, n * 8, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(vals, vb, n * 4, hipMemcpyDeviceToHost) == hipSuccess)
        rc = 0;
    hipFree(ka); hipFree(kb); hipFree(va); hipFree(vb); hipFree(temp);
    return rc;
}
extern "C" int pt_debug_exclusive_scan(uint32_t* data, size_t n) {
    if (n == 0) return 0;
    uint32_t *a = nullptr, *b = nullptr; void* temp = nullptr;
    int rc = -3;
    if (hipMalloc(&a, n * 4) == hipSuccess && hipMalloc(&b, n * 4) == hipSuccess && hipMalloc(&temp, pt::exclusive_scan_temp_bytes(n)) == hipSuccess &&
        hipMemcpy(a, data, n * 4, hipMemcpyHostToDevice) == hipSuccess && pt::exclusive_scan_u32(temp, a, b, n, nullptr) == hipSuccess &&
        hipDeviceSynchronize() == hipSuccess && hipMemcpy(data, b, n * 4, hipMemcpyDeviceToHost) == hipSuccess)
        rc = 0;
    hipFree(a); hipFree(b); hipFree(temp);
    return rc;
}
