// pt_wavefront.hip -- staged (wavefront) arrangement of the path tracer for gfx950.
//
// The frame advances one path vertex at a time through three stages -- trace (closest hit) -> shade ->
// trace (shadow) -- over SoA ray / hit / path-state arrays in HBM (coalesced 16-B-per-lane float4 records).
// The trace stages are small-register kernels that run at 6 waves/SIMD to hide the dependent-load latency
// of BVH traversal; the heavy material code runs only on lanes that have a hit.  MI355X has HBM bandwidth
// to spare (the state traffic is ~300 B per vertex against 8 TB/s) but no RT cores, so occupancy and full
// waves are what buy ray throughput.
//
// Queues are SHARDED: kShards self-contained sub-pipelines.  Workgroup b of every stage launch belongs to
// shard b % kShards; it consumes only its shard's queue segment and pushes only into its shard's segments,
// so an entry never migrates and each segment's size is bounded by what the generate stage put there.
// Surviving paths are compacted with a wave64 ballot and ONE atomic per wave on the shard's counter --
// 32 waves per counter instead of 32 k on one word (a single word saturates at ~88 atomics/us on this
// chip, which made an unsharded queue the bottleneck of every stage).
//
// Determinism: a pixel's radiance is accumulated in its own slot in a fixed order (hit terms, then the
// env-shadow term, then the light-shadow term of that vertex), independent of queue order, so frames are
// bit-reproducible and N tile shards compose bit-exactly.
#define PT_LUT_LDS 1          // every stage kernel of this file stages the sRGB table into LDS (pt_shading.h stage_luts)
#include "pt_vertex.h"
#include "pt_host.h"

#ifdef PT_TIMING                 // diagnostic build only (tools/shade_sections.py); not part of the C-ABI
namespace pt { __device__ unsigned long long pt_timing[12]; }
extern "C" int pt_debug_read_timing(unsigned long long* out12, int reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out12, HIP_SYMBOL(pt::pt_timing), 96);
    if (reset) { unsigned long long z[12] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(pt::pt_timing), z, 96); }
    return 0;
}
#endif
#ifdef PT_UTIL_PROBE             // diagnostic build only (tools/util_probe.py): wave-iterations and active lanes of the traversal phases
namespace pt { __device__ unsigned long long pt_util[16]; }
extern "C" int pt_debug_read_util(unsigned long long* out16, int reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out16, HIP_SYMBOL(pt::pt_util), 128);
    if (reset) { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(pt::pt_util), z, 128); }
    return 0;
}
#define PT_UTIL(i, v) u_acc[i] += (v)
#else
#define PT_UTIL(i, v)
#endif
namespace pt {

constexpr uint32_t kShards = 256;
constexpr uint32_t kMissTri = 0x7fffffffu;
constexpr uint32_t kNoHint = 0xffffffffu;
#ifndef PT_OCC_CACHE
#define PT_OCC_CACHE 0        // 1: occluder cache for the shadow rays of first path vertices (WfBuffers::occ_cache).  MEASURED (round 3, images
                              // bit-identical): node visits per ray 12.6 -> 10.9, traversal 13.85 -> 13.6 ms, shade +1.5 % (the hint fetches):
                              // no gain -- the rays it shortens are the coherent, cheap ones, and a wave still lasts as long as its longest ray.
                              // Off; kept as a build option for the record (profiles/EXPERIMENTS.md).
#endif
constexpr uint32_t kCounterStride = 16;      // one 64-B line per shard counter
constexpr int kCounterArrays = 7;

struct WfBuffers {
    // per slot (one path per pixel of this rank)
    float4* L;            // xyz radiance so far; w = bits: bit 0 env term pending, bit 1 light term pending
    float4* beta_pdf;     // beta.xyz, prev_pdf
    float4* thr_misc;     // thr.xyz, bits: rc | bounce << 16 | prev_mis << 31
    float4* pend;         // [2 * slot] xyz pending env-NEE term (beta-weighted), w = its shadow transmission (written by the shadow stage);
                          // [2 * slot + 1] the same for the punctual-light term: one 32-B piece per path
    // closest-ray queues (ping-pong), kShards segments of seg_cap entries: (o.xyz, tmax), (d.xyz, slot)
    float4* ray_o[2];
    float4* ray_d[2];
    float4* q_beta[2];    // PT_BT_IN_QUEUE: the path's (beta, prev_pdf) and (throughput, misc bits) travel with its closest-ray entry -- written
    float4* q_thr[2];     // compacted, read coalesced -- instead of living in the slot-indexed arrays beta_pdf / thr_misc above
    float4* hit;          // per entry of the current queue: t, u, v, bits: tri | front << 31 (kMissTri: miss)
    float4* env_a;        // per entry of the current closest queue: the vertex's environment light sample, drawn by the traversal stage that
    float4* env_b;        // traces the entry (env_prepass): (direction, pdf), (radiance, -)
    // shadow queue, kShards segments of 2 * seg_cap entries: (o.xyz, bits: slot | is_light << 31), (d.xyz, tmax)
    float4* sh_o;
    float4* sh_d;         // (d.xyz, bits: occluder hint -- a triangle index, kNoHint for none; the rays' tmax is the constant max_ray_length)
    uint32_t* sh_c;       // per shadow entry: where the ray's occluder goes in occ_cache (kNoHint: nowhere)
    // Occluder cache (PT_OCC_CACHE): per pixel, eight triangle indices -- [0] the last triangle that occluded an environment shadow ray of the
    // pixel's FIRST path vertex, [1 + light % 7] the same for that punctual light.  A hint, never a result: the shadow ray tests the hinted
    // triangle first, with the same intersection routine and flags; if it hits, the ray is occluded exactly as the traversal would have found,
    // by whichever triangle.  Persists across pt_trace calls (the next samples of a pixel meet the occluders of the last ones).
    uint32_t* occ_cache;
    uint32_t* cnt[7];     // per shard (stride kCounterStride): entry counts of closest queue 0, closest queue 1, shadow queue (even bounces);
                          // [3], [4]: dynamic-fetch heads of the closest / shadow trace stages; [5]: shadow-queue count of odd bounces
                          // (the shadow count ping-pongs so that the fused traversal stage can zero the one the NEXT shade stage fills
                          // while it still reads the current one); [6]: dynamic-fetch head of the shade stage
    uint32_t capacity;    // slots
    uint32_t chunks_per_shard;   // path-state arrays: 256-slot chunks per shard (state_index)
    uint32_t seg_cap;     // entries per closest-queue segment
    uint32_t blocks_per_shard;
    uint32_t gen_region_tiles, gen_rounds;     // k_wf_generate: tiles per XCD band, workgroup-rounds to cover a band's (tile, sample) pairs
};


// Queue entries and path-state records are written once and read once, a stage apart, by then long evicted from the 4-MiB L2s: they
// are moved with the non-temporal hint so that they do not push the tree's nodes and triangle packets out on their way through
// (PT_STREAM: 0 plain, 1 the ray / hit / shadow queues, 2 the per-path state records too).
#ifndef PT_STREAM
#define PT_STREAM 1       // (measured 0 / 1 / 2, Sponza class: 5073 / 5100 / 5099 Mrays/s at 8 spp, 4.66 / 4.59 / 4.57 ms per 1-spp launch)
#endif
typedef float nt_v4f __attribute__((ext_vector_type(4)));
PT_DEV float4 nt_load(const float4& p) { const nt_v4f v = __builtin_nontemporal_load((const nt_v4f*)&p); return make_float4(v.x, v.y, v.z, v.w); }
PT_DEV uint32_t nt_load(const uint32_t& p) { return __builtin_nontemporal_load(&p); }
PT_DEV void nt_store(float4& p, const float4 v) { nt_v4f q; q.x = v.x; q.y = v.y; q.z = v.z; q.w = v.w; __builtin_nontemporal_store(q, (nt_v4f*)&p); }
PT_DEV void nt_store(uint32_t& p, const uint32_t v) { __builtin_nontemporal_store(v, &p); }
#if PT_STREAM >= 1
#define QLD(p) nt_load(p)
#define QST(p, v) nt_store((p), (v))
#else
#define QLD(p) (p)
#define QST(p, v) ((p) = (v))
#endif
#if PT_STREAM >= 2
#define SLD(p) nt_load(p)
#define SST(p, v) nt_store((p), (v))
#else
#define SLD(p) (p)
#define SST(p, v) ((p) = (v))
#endif

// Where the state of path `slot` lives.  Generate workgroup b takes the 256-slot chunks {b, b + grid, ...} and belongs to shard b % 256,
// so in slot order a shard's paths are 4-KB pieces 1 MB apart in each of the six state arrays, and 33 MB apart from sample to sample:
// a shade workgroup touched hundreds of pages (0.7 UTCL1 misses per hit, profiles/r02e_memside_counters.txt).  PT_STATE_BY_SHARD stores
// chunk c of shard s at (s * chunks_per_shard + c / 256): everything a shard reads and writes is one contiguous piece of each array.
#ifndef PT_STATE_BY_SHARD
#define PT_STATE_BY_SHARD 0      // (measured on MI355X: 5082 against 5081 Mrays/s, no effect -- kept as a build option)
#endif
PT_DEV uint32_t state_index(const WfBuffers& wf, uint32_t slot) {
#if PT_STATE_BY_SHARD
    const uint32_t c = slot >> 8;
    return (((c & (kShards - 1u)) * wf.chunks_per_shard + (c >> 8)) << 8) | (slot & 255u);
#else
    return slot;
#endif
}
#define SIDX(s) state_index(wf, (s))
#define PEND_ENV(s) wf.pend[2u * SIDX(s)]
#define PEND_LIGHT(s) wf.pend[2u * SIDX(s) + 1u]
// The random-sequence counter every path holds after its camera ray (camera_ray draws once): the state of a path at its FIRST vertex is a
// constant -- L = 0, beta = 1, pdf = 0, throughput = 1, rc = kRcAfterCamera, nothing pending -- so the generate stage writes no state and
// the first shade stage reads none (PT_FIRST_VERTEX_STATELESS).
constexpr int kRcAfterCamera = 1;
// The shade stage is bound by the divergent vector-memory instructions it issues (tools/pmc_shade_attribution.sh: taking the radiance /
// pending records away -- 3 loads, 3 stores, 13 % of its fabric bytes -- made it 10 % faster).  beta / throughput are read by exactly one
// consumer, the shade stage of the next vertex, which already reads the path's queue entry: with PT_BT_IN_QUEUE they ride in two more
// arrays parallel to the closest-ray queue (coalesced both ways) instead of two slot-indexed arrays (a divergent load and store each).
#ifndef PT_BT_IN_QUEUE
#define PT_BT_IN_QUEUE 1
#endif
#ifndef PT_FIRST_VERTEX_STATELESS
#define PT_FIRST_VERTEX_STATELESS 1
#endif

// wave64 ballot compaction into a shard counter: lanes with `pred` get consecutive indices; one atomic per wave.
PT_DEV uint32_t queue_push(uint32_t* counter, bool pred) {
    const unsigned long long m = __ballot(pred);
    if (m == 0) return 0;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t rank = __popcll(m & ((1ull << lane) - 1ull));
    uint32_t base = 0;
    const int leader = __ffsll((long long)m) - 1;
    if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = __shfl(base, leader, 64);
    return base + rank;
}

// Every stage launch has the same grid: kShards * blocks_per_shard workgroups.  Workgroup b: shard b % kShards, member b / kShards.
struct ShardView { uint32_t shard, member, stride; };
PT_DEV ShardView shard_view(const WfBuffers& wf) {
    ShardView v;
    v.shard = blockIdx.x % kShards;
    v.member = blockIdx.x / kShards;
    v.stride = wf.blocks_per_shard * kBlock;
    return v;
}

// Which pixel tiles a workgroup generates.  Workgroups are dispatched to the eight XCDs round-robin (XCD = blockIdx % 8), every later
// stage launch has the same grid, and shard s = blockIdx % kShards is only ever touched by workgroups with blockIdx % 8 == s % 8:
// a path lives its whole life on ONE XCD.  Each XCD has its own 4-MiB L2, and the scene (BVH + packets, tens of MB) fits none of
// them.  PT_GEN_XCD_BANDS = 1 / 2 gives each XCD a CONTIGUOUS range of this rank's tiles (a row band / a column band of the
// screen), so that its primary rays, their shadow rays and most first bounces walk one part of the scene and its L2 holds that part
// instead of a 1/8 sample of everything.  Within the band, consecutive workgroups of the XCD take consecutive (tile, sample)
// pairs.  Slots keep their meaning (slot = sample * pixel_slots + tile * 256 + lane), only the workgroup that generates a slot
// changes: images are bit-identical (tested).  MEASURED: both band shapes are 3-4 % SLOWER than dealing tiles round-robin over all
// workgroups (the default, 0): a path never leaves its XCD, so the XCD whose band holds the expensive part of the picture finishes
// last while the others idle, and that costs more than the locality buys.  Kept as a build option for the record.
constexpr uint32_t kXcds = 8;
#ifndef PT_GEN_XCD_BANDS
#define PT_GEN_XCD_BANDS 0      // measured on MI355X (Sponza-class 1080p, 8 spp / launch): 0 (tiles dealt round-robin) 4473 Mrays/s,
                                // 1 (row bands) 4339, 2 (column bands) 4274 -- see the comment above and DESIGN.md section 4
#endif
__global__ __launch_bounds__(kBlock) void k_wf_generate(FrameConstants fc, WfBuffers wf, Counters* __restrict__ counters) {
    const ShardView sv = shard_view(wf);
    const uint32_t per_xcd = gridDim.x / kXcds;                       // the grid is a multiple of kShards, kShards of kXcds
    const uint32_t xcd = blockIdx.x % kXcds, member = blockIdx.x / kXcds;
    uint32_t first_tile = xcd * wf.gen_region_tiles;
    uint32_t n_tiles = first_tile < fc.my_tiles ? min(wf.gen_region_tiles, fc.my_tiles - first_tile) : 0u;
#if PT_GEN_XCD_BANDS == 2
    // column bands (whole frames only; a tile shard of a frame keeps the ranges of its own tile list): XCD x owns the tile columns
    // [c0, c1) over all rows -- every band holds sky, walls and floor alike, so the eight XCDs finish together
    const bool columns = fc.tile_rank_count == 1;
    const uint32_t c0 = xcd * fc.tiles_x / kXcds, band_w = (xcd + 1u) * fc.tiles_x / kXcds - c0;
    if (columns) n_tiles = band_w * fc.tiles_y;
#endif
    unsigned n_primary = 0;
    for (uint32_t rnd = 0; rnd < wf.gen_rounds; rnd++) {
        // a workgroup-round is one 16x16 tile of one sample: primary rays stay coherent per wave (8x8 quadrant)
#if PT_GEN_XCD_BANDS
        const uint32_t q = rnd * per_xcd + member;
        const uint32_t tile_in_region = q / fc.spp, sample = q - tile_in_region * fc.spp;
        uint32_t tile = first_tile + tile_in_region;
#if PT_GEN_XCD_BANDS == 2
        if (columns) { const uint32_t brow = band_w ? tile_in_region / band_w : 0u; tile = brow * fc.tiles_x + c0 + (tile_in_region - brow * band_w); }
#endif
        const uint32_t slot = sample * fc.pixel_slots + tile * kBlock + threadIdx.x;
        uint32_t px = 0, py = 0;
        const bool valid = tile_in_region < n_tiles && slot_pixel(fc, slot, px, py);
#else       // A/B: tiles dealt round-robin over all workgroups (every XCD sees the whole screen); rounds are sized for either
        const uint32_t slot = (rnd * gridDim.x + blockIdx.x) * kBlock + threadIdx.x;
        const uint32_t sample = slot_sample(fc, slot);
        uint32_t px = 0, py = 0;
        const bool valid = slot < wf.capacity && slot_pixel(fc, slot, px, py);
#endif
        int rc = 0;
        Ray ray;
        ray.o = v3(0); ray.d = v3(0, 0, 1); ray.tmin = 0; ray.tmax = 0;
        if (valid) ray = camera_ray(fc, sample_seed(fc, sample), px, py, rc);
        const uint32_t idx = queue_push(wf.cnt[0] + sv.shard * kCounterStride, valid);
        if (valid) {
            const size_t e = (size_t)sv.shard * wf.seg_cap + idx;
            QST(wf.ray_o[0][e], make_float4(ray.o.x, ray.o.y, ray.o.z, ray.tmax));
            QST(wf.ray_d[0][e], make_float4(ray.d.x, ray.d.y, ray.d.z, __uint_as_float(slot)));
#if !PT_FIRST_VERTEX_STATELESS
            SST(wf.L[SIDX(slot)], make_float4(0, 0, 0, 0));
#if PT_BT_IN_QUEUE
            QST(wf.q_beta[0][e], make_float4(1, 1, 1, 0));
            QST(wf.q_thr[0][e], make_float4(1, 1, 1, __uint_as_float((uint32_t)rc)));
#else
            SST(wf.beta_pdf[SIDX(slot)], make_float4(1, 1, 1, 0));
            SST(wf.thr_misc[SIDX(slot)], make_float4(1, 1, 1, __uint_as_float((uint32_t)rc)));
#endif
#endif
            n_primary++;
        }
    }
    LaneStats st = {0, 0, 0, 0};
    flush_counters(counters, threadIdx.x & 63, n_primary, 0, 0, 0, st);
}

// Wave-persistent "while-while" traversal of one shard segment with dynamic ray fetch (pt_traverse.h).
// MODE 0: closest hits of queue `cur` -> wf.hit.  MODE 1: occlusion of the shadow queue -> pend_*.w.
#ifndef PT_REFILL
#define PT_REFILL 32          // idle lanes that trigger a refill from the shard queue (swept 8..64 on MI355X: 48 was best in round 1; re-swept after the sample pre-pass: 16 / 24 / 32 / 40 / 48 / 56 -> 5470 / 5507 / 5538 / 5536 / 5500 / 5275 Mrays/s)
#endif
PT_DEV int shadow_counter(int bounce) { return (bounce & 1) ? 5 : 2; }
template <bool COUNT, int MODE>
PT_DEV void trace_persistent(const SceneRec& sc, const WfBuffers& wf, int* my_stack, const ShardView& sv, int cur, uint32_t rf_closest, uint32_t rmask,
                             uint32_t flags, LaneStats& st, float shadow_tmax = 0.0f, bool use_occ_cache = false) {
    // MODE 0: `cur` = closest queue (0 / 1).  MODE 1: `cur` = index of the shadow-queue counter (shadow_counter(bounce)).
    const uint32_t n = wf.cnt[cur][sv.shard * kCounterStride];
    uint32_t* head = wf.cnt[MODE == 0 ? 3 : 4] + sv.shard * kCounterStride;
    const size_t base = MODE == 0 ? (size_t)sv.shard * wf.seg_cap : (size_t)sv.shard * wf.seg_cap * 2;
    const uint32_t lane = threadIdx.x & 63;
    int spill[kStackSpill];
    Trav t;
    t.cur = kTravDone; t.sp = 0;
    bool has = false, exhausted = (n == 0);
    uint32_t entry = 0, slot_bits = 0, occ_at = kNoHint;
#ifdef PT_UTIL_PROBE
    unsigned long long u_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // wave-level: [0] node iterations [1] lanes stepping [2] leaf iterations [3] lanes testing [4] refills [5] lanes refilled [6] lanes holding a ray, summed over node iterations
#endif
    for (;;) {
        // ---- refill idle lanes from the shard queue: ballot + one atomic per wave
        const unsigned long long idle = __ballot(!has);
        const uint32_t nidle = (uint32_t)__popcll(idle);
        if (!exhausted && nidle >= PT_REFILL) {
            const int leader = __ffsll((long long)idle) - 1;
            uint32_t first = 0;
            if ((int)lane == leader) first = atomicAdd(head, nidle);
            first = __shfl(first, leader, 64);
            PT_UTIL(4, 1); PT_UTIL(5, min(nidle, first < n ? n - first : 0u));
            if (!has) {
                const uint32_t i = first + (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                if (i < n) {
                    Ray ray;
                    if (MODE == 0) {
                        const float4 o = QLD(wf.ray_o[cur][base + i]), d = QLD(wf.ray_d[cur][base + i]);
                        ray.o = v3(o.x, o.y, o.z); ray.tmin = 0; ray.d = v3(d.x, d.y, d.z); ray.tmax = o.w;
                        trav_init(t, sc, ray, rf_closest, rmask, 0, 0.0f);
                    } else {
                        const float4 o = QLD(wf.sh_o[base + i]), d = QLD(wf.sh_d[base + i]);
                        slot_bits = __float_as_uint(o.w);
                        ray.o = v3(o.x, o.y, o.z); ray.tmin = 0; ray.d = v3(d.x, d.y, d.z); ray.tmax = shadow_tmax;
                        const bool alpha_shadow = (slot_bits >> 31) && (flags & PT_FLAG_ALPHA_SHADOWS);          // TraceShadowRay :724-742
                        uint32_t srf = (flags & PT_FLAG_CULL_BACKFACE) ? RF_CULL_BACK : 0;
                        srf |= alpha_shadow ? RF_FORCE_NON_OPAQUE : RF_ACCEPT_FIRST;
                        trav_init(t, sc, ray, srf, 0xff, 1, alpha_shadow ? 1.0f : 0.0f);
#if PT_OCC_CACHE
                        // occluder cache: an accept-first ray whose pixel remembers an occluder tests that triangle FIRST -- as a one-triangle leaf,
                        // with the root waiting on the stack -- and ends there if it still occludes
                        occ_at = kNoHint;
                        if (use_occ_cache) {
                            const uint32_t hint = __float_as_uint(d.w);
                            occ_at = alpha_shadow ? kNoHint : QLD(wf.sh_c[base + i]);
                            if (occ_at != kNoHint && hint < sc.num_tris && hint < kLeafFirstMask && t.cur >= 0) { my_stack[0] = t.cur; t.sp = 1; t.cur = ~(int)hint; }
                        }
#endif
                    }
                    entry = i;
                    has = true;
                }
            }
            if (first + nidle >= n) exhausted = true;
        }
        if (__ballot(has) == 0) {
            if (exhausted) break;
            continue;                                       // cannot happen (all idle -> nidle = 64 >= PT_REFILL), kept for safety
        }
        // ---- node phase: every lane that holds an inner node steps until none does
#ifndef PT_NODE_MIN
#define PT_NODE_MIN 16        // (swept 4..32 at 8 samples per launch: 16) leave the node phase early when fewer lanes than this hold an inner node AND some lane waits at a leaf
                              // (swept 1/4/8/12/16/32 on MI355X: 2195/2326/2359/2360/2356/2294 Mrays/s)
#endif
#ifndef PT_POSTPONE_LEAF
#define PT_POSTPONE_LEAF 0    // 1: a lane that reaches a leaf while others still step nodes sets the leaf aside and goes on with its next stack entry.
                              // MEASURED (round 3, bit-identical images): lanes stepping per node iteration 42.6 -> 44.5 of 64, but 25 % more node
                              // visits (12.6 -> 15.8 per ray; occlusion rays +36 %: the postponed leaf is usually the hit that ends them) --
                              // traversal 13.9 -> 16.1 ms per 8-spp launch.  Off; kept as a build option for the record.
#endif
        for (;;) {
#if PT_POSTPONE_LEAF
            // Speculative traversal (Aila & Laine): a third of the lanes of a node iteration used to sit at a leaf, waiting for the wave's leaf
            // phase (tools/util_probe.py: 42.6 of 59 lanes stepping).  Such a lane now POSTPONES its leaf -- one at a time, and only if its
            // stack holds something to go on with -- and keeps stepping; the leaf phase tests the postponed leaf FIRST, so triangles are
            // tested in the order they were reached and every ray finds the hit it found before.  The node steps taken meanwhile used the
            // older, longer interval: a superset of the nodes, never a different hit.
            if (has && t.cur < 0 && t.cur != kTravDone && t.post == kTravDone && t.sp > 0) { t.post = t.cur; trav_pop(t, sc, my_stack, spill); }
#endif
            const unsigned long long at_node = __ballot(has && t.cur >= 0);
            if (at_node == 0) break;
            if (PT_NODE_MIN > 1 && (int)__popcll(at_node) < PT_NODE_MIN && __ballot(has && t.cur < 0 && (t.cur != kTravDone || t.post != kTravDone)) != 0) break;
            PT_UTIL(0, 1); PT_UTIL(1, __popcll(at_node)); PT_UTIL(6, __popcll(__ballot(has)));
            if (has && t.cur >= 0) trav_node_step<COUNT, MODE == 0>(t, sc, my_stack, spill, st);
        }
        // ---- leaf phase
#ifdef PT_UTIL_PROBE
        { const unsigned long long at_leaf = __ballot(has && ((t.cur != kTravDone && t.cur < 0) || t.post != kTravDone)); if (at_leaf) { PT_UTIL(2, 1); PT_UTIL(3, __popcll(at_leaf)); } }
#endif
#if PT_POSTPONE_LEAF
        // the postponed leaf first (the lane then goes on with the entry it had moved on to, unless the ray ended there), then the leaf the
        // lane stands at: ONE copy of the leaf code, run at most twice
#pragma nounroll
        for (int pass = 0; pass < 2; pass++) {
            const bool first_pass = pass == 0;
            const bool mine = has && (first_pass ? t.post != kTravDone : (t.cur != kTravDone && t.cur < 0));
            if (!__any(mine)) continue;
            if (mine) {
                const int keep = t.cur;
                if (first_pass) { t.cur = t.post; t.post = kTravDone; }
                trav_leaf_step<COUNT>(t, sc, my_stack, spill, st, first_pass, keep);
            }
        }
#else
        if (has && t.cur != kTravDone && t.cur < 0) trav_leaf_step<COUNT>(t, sc, my_stack, spill, st);
#endif
        // ---- retire finished rays
        if (has && t.cur == kTravDone) {
            if (MODE == 0) {
                const uint32_t bits = t.committed ? ((uint32_t)t.best.tri | (t.best.front ? 0x80000000u : 0u)) : kMissTri;
                QST(wf.hit[base + entry], make_float4(t.best.t, t.best.u, t.best.v, __uint_as_float(bits)));
            } else {
                const float tr = t.committed ? t.transmission : 1.0f;                                              // ShadowMiss :1081-1085
                const uint32_t slot = slot_bits & 0x7fffffffu;
#if PT_OCC_CACHE
                if (use_occ_cache && occ_at != kNoHint && t.committed) wf.occ_cache[occ_at] = (uint32_t)t.best.tri;   // remember the occluder
#endif
                float* w = (slot_bits >> 31) ? &PEND_LIGHT(slot).w : &PEND_ENV(slot).w;
                *w = tr;
            }
            has = false;
        }
    }
#ifdef PT_UTIL_PROBE
    if (lane == 0) for (int k = 0; k < 7; k++) atomicAdd(&pt_util[MODE * 8 + k], u_acc[k]);
#endif
}

// Environment light samples of the path vertices the closest-hit rays of queue `cur` will reach (SampleEnvironmentMap,
// PathTracer.lib.hlsl:688-703: random numbers -> importance-map descent -> direction -> cube-map radiance).  The sample depends on the
// pixel's random sequence only, not on the hit, so it is drawn HERE, by the small-register traversal kernel at 6 waves per SIMD,
// where its three dependent gathers hide behind other waves, instead of in the shade stage at 2 waves per SIMD, where they were a
// fifth of that stage's time (tools/shade_sections.py).  The shade stage reads the result with the queue entry (coalesced, no extra
// round trip) and still counts the draw.  LDS: the importance pyramid's coarse levels are staged into the traversal stack's memory,
// which is idle until the traversal starts.
// Every traversal workgroup draws its share of the shard's samples before it starts traversing.  Measured per 8-spp launch of the bench
// scene: the shade stage 12.1 -> 10.5 ms, the traversal stages 13.0 -> 14.4 ms -- the ~2 k vector instructions of a sample cost nearly as
// much here as there (+0.8 % overall, and 17 KB of LDS a shade workgroup no longer needs).  Handing the samples to dedicated workgroups
// prepended to the traversal launch (1-3 per shard), so that their arithmetic would fill the issue slots of the memory-bound traversal
// waves beside them, was slower still (traversal 14.9-15.0 ms): those slots are not idle.
#ifndef PT_ENV_PREPASS
#define PT_ENV_PREPASS 1
#endif

PT_DEV bool env_prepass_wanted(const SceneRec& sc, const FrameConstants& fc, int vertex_bounce) {
    return PT_ENV_PREPASS && sc.has_env && (fc.flags & PT_FLAG_ENVIRONMENT_MAP) && (fc.flags & PT_FLAG_ENVIRONMENT_MIS) && vertex_bounce < fc.max_bounces;
}
PT_DEV void env_prepass(const SceneRec& sc, const FrameConstants& fc, const WfBuffers& wf, int* stack_lds, const ShardView& sv, int cur, bool first_vertex) {
    static_assert((size_t)kStackLds * kBlock * sizeof(int) >= (size_t)kImpLdsFloat4 * sizeof(float4), "the traversal stack's LDS must hold the importance pyramid's coarse levels");
    float4* top = (float4*)stack_lds;
    stage_importance_top_into(sc, top);
    const uint32_t n = wf.cnt[cur][sv.shard * kCounterStride];
    const size_t base = (size_t)sv.shard * wf.seg_cap;
    for (uint32_t i = sv.member * kBlock + threadIdx.x; i < n; i += sv.stride) {
        const uint32_t slot = QLD(*((const uint32_t*)&wf.ray_d[cur][base + i] + 3));
        int rc = kRcAfterCamera;
#if PT_BT_IN_QUEUE
        if (!(PT_FIRST_VERTEX_STATELESS && first_vertex)) rc = (int)(QLD(*((const uint32_t*)&wf.q_thr[cur][base + i] + 3)) & 0xffffu);
#else
        if (!(PT_FIRST_VERTEX_STATELESS && first_vertex)) rc = (int)(SLD(*((const uint32_t*)&wf.thr_misc[SIDX(slot)] + 3)) & 0xffffu);
#endif
        uint32_t px, py;
        slot_pixel(fc, slot, px, py);
        const vec4 r = next_random(px, py, sample_seed(fc, slot_sample(fc, slot)), rc);
        const EnvSample e = environment_light_sample(sc, fc.environment_intensity, r.x, r.y, top);
        QST(wf.env_a[base + i], make_float4(e.dir.x, e.dir.y, e.dir.z, e.pdf));
        QST(wf.env_b[base + i], make_float4(e.color.x, e.color.y, e.color.z, 0.0f));
    }
    __syncthreads();                       // the stack memory goes back to the traversal
}

#ifndef PT_TRACE_WAVES
#define PT_TRACE_WAVES 1
#endif
#ifndef PT_LATE_GRID
#define PT_LATE_GRID 1        // smaller stage grids for the thin late bounces (launch_wavefront)
#endif
#ifndef PT_FUSE_TRAVERSAL
#define PT_FUSE_TRAVERSAL 1   // the shadow rays of a bounce and the closest-hit rays of the next in one launch (k_wf_traverse)
#endif
// `bounce` = the bounce whose shade stage follows: it fills closest queue cur ^ 1 and the shadow counter of that bounce.
// DEFAULTS (k_wf_trace, k_wf_traverse): a copy compiled for the settings that leave the rays' flags alone -- no back-face culling, no alpha
// shadows, no indirect-environment-only mask (the application's defaults): ray flags 0, mask 0xff and an accept-first shadow search as
// constants take the other searches' code out of the loop (13.87 against 14.04 ms of traversal per launch).  launch_wavefront picks it.
constexpr uint32_t kTravFlagMask = PT_FLAG_CULL_BACKFACE | PT_FLAG_ALPHA_SHADOWS | PT_FLAG_INDIRECT_ENVIRONMENT_ONLY;
#ifndef PT_TRAV_SPECIALISE
#define PT_TRAV_SPECIALISE 1
#endif
template <bool COUNT, bool DEFAULTS>
__global__ __launch_bounds__(kBlock, PT_TRACE_WAVES) void k_wf_trace(SceneRec sc, FrameConstants fc, WfBuffers wf, int cur, int bounce, uint32_t rf, uint32_t rmask, Counters* __restrict__ counters) {
    if (DEFAULTS) { rf = 0; rmask = 0xff; }
    __shared__ int s_stack[kStackLds * kBlock];
    stage_luts(sc);
    const ShardView sv = shard_view(wf);
    if (env_prepass_wanted(sc, fc, bounce)) env_prepass(sc, fc, wf, s_stack, sv, cur, bounce == 0);
    // member 0 of each shard zeroes the counters the following shade stage fills
    if (sv.member == 0 && threadIdx.x == 0) { wf.cnt[cur ^ 1][sv.shard * kCounterStride] = 0; wf.cnt[shadow_counter(bounce)][sv.shard * kCounterStride] = 0; wf.cnt[6][sv.shard * kCounterStride] = 0; }
    LaneStats st = {0, 0, 0, 0};
    trace_persistent<COUNT, 0>(sc, wf, s_stack + threadIdx.x, sv, cur, rf, rmask, 0, st);
    if (COUNT) { flush_counters(counters, threadIdx.x & 63, 0, 0, 0, 0, st); if (st.deep) atomicAdd(&counters->deep_pushes, (unsigned long long)st.deep); }
    else if (st.overflow | st.deep) flush_rare(counters, st);
}

// Fused traversal stage: the occlusion rays of bounce `bounce` AND the closest-hit rays of bounce + 1.  Both were produced by the
// shade stage of `bounce` and neither needs the other's result (the shadow transmissions are only read by the NEXT shade stage), so
// one launch serves both: a wave that runs out of shadow rays goes straight on to pull bounce rays, and the frame has two
// grid-wide synchronisations per bounce instead of three (each one ends on its slowest wave: ~0.07 ms of a 5.5-ms 1-spp frame).
// `nxt` = closest queue the shade stage of `bounce` filled.  Zeroes what the shade stage of bounce + 1 fills.
template <bool COUNT, bool DEFAULTS>
__global__ __launch_bounds__(kBlock, PT_TRACE_WAVES) void k_wf_traverse(SceneRec sc, FrameConstants fc, WfBuffers wf, int nxt, int bounce, uint32_t rf, uint32_t rmask, uint32_t flags,
                                                                        Counters* __restrict__ counters) {
    if (DEFAULTS) { rf = 0; rmask = 0xff; flags &= ~kTravFlagMask; }
    __shared__ int s_stack[kStackLds * kBlock];
    stage_luts(sc);
    const ShardView sv = shard_view(wf);
    if (env_prepass_wanted(sc, fc, bounce + 1)) env_prepass(sc, fc, wf, s_stack, sv, nxt, false);
    if (sv.member == 0 && threadIdx.x == 0) { wf.cnt[nxt ^ 1][sv.shard * kCounterStride] = 0; wf.cnt[shadow_counter(bounce + 1)][sv.shard * kCounterStride] = 0; wf.cnt[6][sv.shard * kCounterStride] = 0; }
    LaneStats st_shadow = {0, 0, 0, 0}, st = {0, 0, 0, 0};
    trace_persistent<COUNT, 1>(sc, wf, s_stack + threadIdx.x, sv, shadow_counter(bounce), 0, 0xff, flags, st_shadow, fc.max_ray_length, wf.occ_cache != nullptr && bounce == 0);
    trace_persistent<COUNT, 0>(sc, wf, s_stack + threadIdx.x, sv, nxt, rf, rmask, 0, st);
    if (COUNT) {
        flush_counters(counters, threadIdx.x & 63, 0, 0, 0, 0, st_shadow, true); flush_counters(counters, threadIdx.x & 63, 0, 0, 0, 0, st);
        if (st.deep | st_shadow.deep) atomicAdd(&counters->deep_pushes, (unsigned long long)st.deep + st_shadow.deep);
    } else if (st.overflow | st_shadow.overflow | st.deep | st_shadow.deep) { flush_rare(counters, st); flush_rare(counters, st_shadow); }
}

// The reference multiplies the light colour by the shadow transmission BEFORE `if (any(color > 0))` and never evaluates
// the BSDF of an occluded sample: an occluded sample contributes nothing even when its pending term is NaN.
// Both pending records are fetched whatever the flags say (the slots always exist): three loads in one round trip instead of
// the flags first and the records behind them.
// `pf` = the pending bits that travel in L.w.
PT_DEV void apply_pending(const WfBuffers& wf, uint32_t slot, uint32_t pf, vec3& L) {
    const float4 pe = SLD(PEND_ENV(slot)), pl = SLD(PEND_LIGHT(slot));
    if ((pf & 1u) && pe.w > 0.0f) L += v3(pe.x, pe.y, pe.z) * pe.w;
    if ((pf & 2u) && pl.w > 0.0f) L += v3(pl.x, pl.y, pl.z) * pl.w;
}

// Material features whose hits the shade stage sets aside and shades together (k_wf_shade): 1 sheen, 2 clearcoat, 4 transmission, 8 anisotropy;
// 0 (the default): none, the code is not compiled in.  MEASURED with 1 (sheen) on the bench scene, where 3 % of the hits put a sheen lane into
// nearly every wave: shade stage 9.06 -> 8.80 ms per 8-sample launch (+1 % rays/s), but a single-sample launch 4.00 -> 4.05 ms (a wave rarely
// collects enough notes before its queue ends, and the last iteration over the few it has is an iteration more), the material grid -1 %
// (a sixth of its hits are sheen: the extra iterations cost more than the shared code saves) and a scene without sheen -0.8 % (the test
// itself).  The noted hits cost a full iteration per 60 while the lanes they left idle shorten nothing, which eats most of what the
// shared lobe code saves.  Enabled per scene by the host (FrameConstants::defer_rare) when compiled in.  profiles/EXPERIMENTS.md.
#ifndef PT_DEFER_FEATURES
#define PT_DEFER_FEATURES 0
#endif
#ifndef PT_DEFER_FLUSH
#define PT_DEFER_FLUSH 60
#endif
#ifndef PT_DEFER_MAJORITY
#define PT_DEFER_MAJORITY 4
#endif
constexpr uint32_t kNoHeld = 0xffffffffu;
constexpr int kDeferFlush = PT_DEFER_FLUSH;        // notes in a wave that trigger the iteration over them
constexpr uint32_t kDeferMajority = PT_DEFER_MAJORITY; // a chunk with this many rare hits is shaded in place
PT_DEV bool material_is_rare(const SceneRec& sc, uint32_t inst_id) {
    const ShadeInst inst = load_shade_inst(sc, inst_id);
    const MatHeader mh = material_header(sc, inst.material_id);
    bool rare = false;
    if (PT_DEFER_FEATURES & 1) rare = rare || mh.sheen_color_factor.x != 0.0f || mh.sheen_color_factor.y != 0.0f || mh.sheen_color_factor.z != 0.0f;
    if (PT_DEFER_FEATURES & 2) rare = rare || mh.clearcoat_factor != 0.0f;
    if (PT_DEFER_FEATURES & 4) rare = rare || mh.transmission_factor != 0.0f;
    if (PT_DEFER_FEATURES & 8) rare = rare || mh.anisotropy_strength != 0.0f;
    return rare;
}

#ifndef PT_SHADE_DYNAMIC
#define PT_SHADE_DYNAMIC 1    // shade-stage waves pull 64-entry chunks from the shard's head counter (0: static rounds over the grid's stride)
#endif
#ifndef PT_SHADE_WAVES
#define PT_SHADE_WAVES 2      // waves per SIMD the register allocator must leave room for (2 -> <= 256 VGPR+AGPR)
#endif
// SPECIAL != 0: a copy of the kernel compiled for ONE setting of the flags the shade stage reads (kShadeFlagMask) and no debug output -- the
// application's defaults (Main.cpp:462-469), with and without punctual lights.  The flag tests are wave-uniform branches either way; as
// constants they also take the code of the other settings (the non-MIS evaluation, the white-material override, 27 debug outputs ...) out of
// the function, and what that does to its register allocation is worth more than the branches: shade stage 9.02 -> 8.69 ms per launch, +1.6 %
// rays/s, images bit-identical.  launch_wavefront picks the copy whose bits match the frame's flags, else the general kernel (SPECIAL = 0).
constexpr uint32_t kShadeFlagMask = PT_FLAG_MATERIAL_DIFFUSE_WHITE | PT_FLAG_MATERIAL_MIS | PT_FLAG_MATERIAL_USE_GEOMETRIC_NORMALS | PT_FLAG_SHADING_NORMAL_ADAPTATION |
                                    PT_FLAG_ENVIRONMENT_MAP | PT_FLAG_ENVIRONMENT_MIS | PT_FLAG_INDIRECT_ENVIRONMENT_ONLY | PT_FLAG_POINT_LIGHTS | PT_FLAG_SHADOW_RAYS;
constexpr uint32_t kShadeSpecialised = 0x80000000u;          // marks a non-zero SPECIAL (a flag set could be 0)
constexpr uint32_t kShadeDefaultsNoLights = PT_FLAG_SHADOW_RAYS | PT_FLAG_ENVIRONMENT_MAP | PT_FLAG_ENVIRONMENT_MIS | PT_FLAG_MATERIAL_MIS | PT_FLAG_SHADING_NORMAL_ADAPTATION;
constexpr uint32_t kShadeDefaults = kShadeDefaultsNoLights | PT_FLAG_POINT_LIGHTS;
#ifndef PT_SHADE_SPECIALISE
#define PT_SHADE_SPECIALISE 1
#endif
// SMALL: the scene's instance rows, materials and lights all fit the LDS copies (<= 128 / 96 / 32: every BASELINE config): as a constant this
// removes the global-memory branch of every table lookup (shade stage 8.70 -> 8.52 ms per launch).
template <uint32_t SPECIAL, bool SMALL>
__global__ __launch_bounds__(kBlock, PT_SHADE_WAVES) void k_wf_shade(SceneRec sc_in, FrameConstants fc_in, WfBuffers wf, int cur, int bounce, Counters* __restrict__ counters) {
    SceneRec sc = sc_in; sc.small_tables = SMALL ? 1u : 0u;
    FrameConstants fc = fc_in;
    if (SPECIAL) { fc.flags = (fc_in.flags & ~kShadeFlagMask) | (SPECIAL & kShadeFlagMask); fc.debug_output = PT_DEBUG_OUTPUT_NONE; }
    {   // a workgroup whose share of the shard's queue is empty (most of them from the third bounce on) leaves before it stages
        // 67 KB of tables into LDS; member 0 stays for the head rewind below
        const ShardView sv0 = shard_view(wf);
        if (sv0.member != 0 && sv0.member * kBlock >= wf.cnt[cur][sv0.shard * kCounterStride]) return;
    }
    stage_luts(sc);
    stage_tangent_lut(sc);
#if !PT_ENV_PREPASS
    stage_importance_top(sc);
#endif
    stage_lights(sc, fc.num_of_lights);
    stage_instances(sc);
    stage_materials(sc);
    const ShardView sv = shard_view(wf);
    const uint32_t n = wf.cnt[cur][sv.shard * kCounterStride];
    const int nxt = cur ^ 1;
    const size_t base = (size_t)sv.shard * wf.seg_cap, sbase = (size_t)sv.shard * wf.seg_cap * 2;
    uint32_t* cnt_next = wf.cnt[nxt] + sv.shard * kCounterStride;
    uint32_t* cnt_shadow = wf.cnt[shadow_counter(bounce)] + sv.shard * kCounterStride;
    // the dynamic-fetch heads of both trace stages are idle while shading runs: rewind them here
    if (sv.member == 0 && threadIdx.x == 0) { wf.cnt[3][sv.shard * kCounterStride] = 0; wf.cnt[4][sv.shard * kCounterStride] = 0; }
    unsigned n_bounce = 0, n_shadow = 0, n_hits = 0;
    LaneStats st = {0, 0, 0, 0};
#if PT_SHADE_DYNAMIC
    // Every WAVE pulls the next 64 entries of its shard's queue from the shard's head counter (one atomic per wave and chunk, the next
    // chunk requested before the current one is shaded, so its round trip is hidden).  Hits differ in cost and a queue is rarely a
    // multiple of the grid's stride: with static rounds a launch ran as long as the workgroups that had one chunk more (a 1-spp
    // 1080p frame: 5.27 rounds' worth of work took 6 rounds).
    uint32_t* shade_head = wf.cnt[6] + sv.shard * kCounterStride;
    const uint32_t lane64 = threadIdx.x & 63u;
    uint32_t chunk = 0;
    if (lane64 == 0) chunk = atomicAdd(shade_head, 64u);
    chunk = (uint32_t)__builtin_amdgcn_readfirstlane((int)chunk);
    // Hits on RARE materials (PT_DEFER_FEATURES: sheen, ...) are set aside and shaded together.  A wave executes a lobe's code if ONE of its
    // lanes needs it, and 3 % of the bench scene's hits (its curtains) put a sheen lane into nearly every wave: the lobe's ~600 instructions
    // a hit ran in almost every wave with two or three lanes active.  Instead, a lane that meets a rare hit only notes the queue entry
    // (`held`: one note per lane, handed to any free lane of the wave through LDS so that the notes pack densely) and does not shade it;
    // once most lanes hold a note -- or the queue is empty -- the wave shades the noted entries in one iteration, all lanes in the rare
    // code together.  Only the order of the follow-up queues changes: images are bit-identical (tools/compare_builds.py).
    const bool defer_on = PT_DEFER_FEATURES != 0 && fc.defer_rare != 0;
    uint32_t held = kNoHeld;
    __shared__ uint32_t s_hand[kBlock];                              // the hand-over slots, 64 per wave
    uint32_t* hand = s_hand + (threadIdx.x & ~63u);
    for (;;) {
        const bool more = chunk < n;
        const unsigned long long held_mask = __ballot(held != kNoHeld);
        if (!more && held_mask == 0) break;
        const bool flush = defer_on && (!more || __popcll(held_mask) >= kDeferFlush);       // (wave-uniform)
        uint32_t next_chunk = 0;
        if (!flush && lane64 == 0) next_chunk = atomicAdd(shade_head, 64u);
        const uint32_t i = flush ? held : chunk + lane64;
        const bool active = flush ? held != kNoHeld : i < n;
        if (flush) held = kNoHeld;
#else
    const bool flush = true;                                          // (no setting aside in this mode)
    const unsigned long long held_mask = 0; uint32_t held = kNoHeld; uint32_t* hand = nullptr; const uint32_t lane64 = threadIdx.x & 63u;
    const uint32_t rounds = (n + sv.stride - 1) / sv.stride;      // uniform per workgroup: ballots inside stay wave-uniform
    for (uint32_t rnd = 0; rnd < rounds; rnd++) {
        const uint32_t i = rnd * sv.stride + sv.member * kBlock + threadIdx.x;
        const bool active = i < n;
#endif
        bool push_env = false, push_light = false, push_bounce = false;
        uint32_t occ_pixel = kNoHint;                       // first-vertex hits: the pixel's row of the occluder cache
        Followups fu;
        fu.q_env = fu.q_light = fu.q_bounce = false;
        uint32_t slot = 0;
        PathState ps;
        ps.beta = v3(0); ps.thr = v3(0); ps.prev_pdf = 0; ps.rc = 0; ps.bounce = 0; ps.prev_mis = false;
        // ---- 1. everything the entry needs, fetched in one round trip
        float4 o = make_float4(0, 0, 0, 0), d = o, h = o, bp = o, tm = o, Lq = o, pe = o, pl = o;
        uint32_t hb = kMissTri;
        const ShadePacket* packet_at = sc.shade;
        RawPacket packet;
#if PT_ENV_PREPASS
        EnvSample es;
        es.dir = v3(0, 0, 1); es.pdf = 1; es.color = v3(0);
#endif
        const bool with_state = !(PT_FIRST_VERTEX_STATELESS && bounce == 0);          // (wave-uniform: `bounce` is a kernel argument)
        if (active) {
            o = QLD(wf.ray_o[cur][base + i]); d = QLD(wf.ray_d[cur][base + i]); h = QLD(wf.hit[base + i]);
#if PT_ENV_PREPASS
            // the vertex's environment light sample, drawn by the traversal stage (env_prepass); without an environment the sample is
            // the constant the in-place code produces.  Fetched with the entry whether or not this vertex will use it: no extra round trip.
            if (env_prepass_wanted(sc, fc, bounce)) {
                const float4 ea = QLD(wf.env_a[base + i]), eb = QLD(wf.env_b[base + i]);
                es.dir = v3(ea.x, ea.y, ea.z); es.pdf = ea.w; es.color = v3(eb.x, eb.y, eb.z);
            }
#endif
            slot = __float_as_uint(d.w);
            // the hit's shading packet depends on the queue entry only, like the path state below: one round trip for both
            hb = __float_as_uint(h.w);
#ifdef PT_PROBE_NO_PACKET     // PROBE ONLY: every hit reads one of 64 packets -- wrong geometry, what the shading-packet gathers cost
            packet_at = sc.shade + (hb == kMissTri ? 0u : (hb & 63u));
#else
            packet_at = sc.shade + (hb == kMissTri ? 0u : (hb & 0x7fffffffu));
#endif
            packet = load_shade_packet_raw(packet_at);
            if (with_state) {
#if PT_BT_IN_QUEUE
                bp = QLD(wf.q_beta[cur][base + i]); tm = QLD(wf.q_thr[cur][base + i]);
#else
                bp = SLD(wf.beta_pdf[SIDX(slot)]); tm = SLD(wf.thr_misc[SIDX(slot)]);
#endif
#ifndef PT_PROBE_NO_LP        // PROBE ONLY: the radiance and pending-term records are neither read nor written -- black image, same paths: what that class of state costs
                Lq = SLD(wf.L[SIDX(slot)]); pe = SLD(PEND_ENV(slot)); pl = SLD(PEND_LIGHT(slot));
#endif
            }
        }
        // ---- 2. rare hits are set aside (wave-uniform control flow)
        bool set_aside = false;
#if PT_SHADE_DYNAMIC
        if (defer_on && !flush) {
            const bool rare = active && hb != kMissTri && material_is_rare(sc, raw_packet_inst(packet));
            const unsigned long long R = __ballot(rare), F = ~held_mask;
            const uint32_t nR = (uint32_t)__popcll(R), nF = (uint32_t)__popcll(F);
            if (nR != 0 && nR < kDeferMajority) {                 // (a chunk made mostly of rare hits is shaded as it is: nothing to gain)
                const unsigned long long below = (1ull << lane64) - 1ull;
                const uint32_t rr = (uint32_t)__popcll(R & below), rf = (uint32_t)__popcll(F & below);
                if (rare && rr < nF) { hand[rr] = i; set_aside = true; }       // the k-th rare lane's note goes to the k-th free lane
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (held == kNoHeld && rf < min(nR, nF)) held = hand[rf];
            }
        }
#endif
        // ---- 3. shade
        if (active && !set_aside) {
            Ray ray;
            ray.o = v3(o.x, o.y, o.z); ray.tmin = 0; ray.d = v3(d.x, d.y, d.z); ray.tmax = o.w;
            vec3 L = v3(0);
            ps.beta = v3(1); ps.prev_pdf = 0; ps.thr = v3(1); ps.rc = kRcAfterCamera; ps.bounce = 0; ps.prev_mis = false;      // a path at its first vertex
            if (with_state) {
                const uint32_t misc = __float_as_uint(tm.w);
                ps.beta = v3(bp.x, bp.y, bp.z); ps.prev_pdf = bp.w; ps.thr = v3(tm.x, tm.y, tm.z);
                ps.rc = (int)(misc & 0xffffu); ps.bounce = (int)((misc >> 16) & 0x7fffu); ps.prev_mis = (misc >> 31) != 0;
#ifndef PT_PROBE_NO_LP
                L = v3(Lq.x, Lq.y, Lq.z);
                const uint32_t pfb = __float_as_uint(Lq.w);                      // the pending bits that travel in L.w
                if ((pfb & 1u) && pe.w > 0.0f) L += v3(pe.x, pe.y, pe.z) * pe.w;
                if ((pfb & 2u) && pl.w > 0.0f) L += v3(pl.x, pl.y, pl.z) * pl.w;
#endif
            }
            uint32_t pf = 0;
            if (hb == kMissTri) L += shade_miss(sc, fc, ray.d, ps);
            else {
                HitRec hit;
                hit.t = h.x; hit.u = h.y; hit.v = h.z; hit.tri = (int)(hb & 0x7fffffffu); hit.front = (hb >> 31) != 0;
                uint32_t px, py;
                slot_pixel(fc, slot, px, py);
                n_hits++;
#if PT_OCC_CACHE
                if (bounce == 0 && wf.occ_cache) occ_pixel = (py * fc.res_x + px) * 8u;
                const uint32_t* occ_row = occ_pixel != kNoHint ? wf.occ_cache + occ_pixel : nullptr;
#else
                const uint32_t* occ_row = nullptr;
#endif
#if PT_ENV_PREPASS
                const bool done = shade_closest_hit<true>(sc, fc, sample_seed(fc, slot_sample(fc, slot)), px, py, ray, hit, packet, packet_at, ps, fu, st.taps, &es, occ_row);
#else
                const bool done = shade_closest_hit(sc, fc, sample_seed(fc, slot_sample(fc, slot)), px, py, ray, hit, packet, packet_at, ps, fu, st.taps, nullptr, occ_row);
#endif
                if (fu.overwrite) L = v3(0);
                L += fu.add;
                n_shadow += fu.counted_shadow;
                if (!done) {
                    push_env = fu.q_env; push_light = fu.q_light; push_bounce = fu.q_bounce;
#ifndef PT_PROBE_NO_LP
                    if (push_env) { pf |= 1u; SST(PEND_ENV(slot), make_float4(fu.pend_env.x, fu.pend_env.y, fu.pend_env.z, 0.0f)); }
                    if (push_light) { pf |= 2u; SST(PEND_LIGHT(slot), make_float4(fu.pend_light.x, fu.pend_light.y, fu.pend_light.z, 0.0f)); }
#endif
                }
            }
#ifndef PT_PROBE_NO_LP
            SST(wf.L[SIDX(slot)], make_float4(L.x, L.y, L.z, __uint_as_float(pf)));
#endif
        }
        // ---- compaction into this shard's shadow segment and next closest-ray segment (wave-uniform control flow)
        // occluder hints of a first-vertex hit's two shadow rays: both cache words of the pixel fetched together (a wave's pixels are an 8x8 block:
        // eight 256-B runs), only by the stage that shades first vertices
        uint32_t at_env = kNoHint, at_light = kNoHint, hint_env = kNoHint, hint_light = kNoHint;
#if PT_OCC_CACHE
        if (bounce == 0 && occ_pixel != kNoHint) {         // (the two cache words were fetched inside shade_closest_hit, well before this point)
            at_env = occ_pixel; at_light = occ_pixel + 1u + fu.light_index % 7u;
            hint_env = fu.hint_env; hint_light = fu.hint_light;
        }
#endif
        const uint32_t ie = queue_push(cnt_shadow, push_env);
        if (push_env) {
            QST(wf.sh_o[sbase + ie], make_float4(fu.origin_above.x, fu.origin_above.y, fu.origin_above.z, __uint_as_float(slot)));
            QST(wf.sh_d[sbase + ie], make_float4(fu.env_dir.x, fu.env_dir.y, fu.env_dir.z, __uint_as_float(hint_env)));
#if PT_OCC_CACHE
            if (bounce == 0 && wf.occ_cache) QST(wf.sh_c[sbase + ie], at_env);
#endif
            n_shadow++;
        }
        const uint32_t il = queue_push(cnt_shadow, push_light);
        if (push_light) {
            QST(wf.sh_o[sbase + il], make_float4(fu.origin_above.x, fu.origin_above.y, fu.origin_above.z, __uint_as_float(slot | 0x80000000u)));
            QST(wf.sh_d[sbase + il], make_float4(fu.light_dir.x, fu.light_dir.y, fu.light_dir.z, __uint_as_float(hint_light)));
#if PT_OCC_CACHE
            if (bounce == 0 && wf.occ_cache) QST(wf.sh_c[sbase + il], at_light);
#endif
            n_shadow++;
        }
        const uint32_t ib = queue_push(cnt_next, push_bounce);
        if (push_bounce) {                                                                           // TraceBounceRay :669-678
            QST(wf.ray_o[nxt][base + ib], make_float4(fu.b_o.x, fu.b_o.y, fu.b_o.z, fc.max_ray_length));
            QST(wf.ray_d[nxt][base + ib], make_float4(fu.b_d.x, fu.b_d.y, fu.b_d.z, __uint_as_float(slot)));
            const uint32_t misc = ((uint32_t)ps.rc & 0xffffu) | ((uint32_t)(ps.bounce + 1) << 16) | (fu.b_mis ? 0x80000000u : 0u);
#if PT_BT_IN_QUEUE
            QST(wf.q_beta[nxt][base + ib], make_float4(fu.b_beta.x, fu.b_beta.y, fu.b_beta.z, fu.b_pdf));
            QST(wf.q_thr[nxt][base + ib], make_float4(fu.b_thr.x, fu.b_thr.y, fu.b_thr.z, __uint_as_float(misc)));
#else
            SST(wf.beta_pdf[SIDX(slot)], make_float4(fu.b_beta.x, fu.b_beta.y, fu.b_beta.z, fu.b_pdf));
            SST(wf.thr_misc[SIDX(slot)], make_float4(fu.b_thr.x, fu.b_thr.y, fu.b_thr.z, __uint_as_float(misc)));
#endif
            n_bounce++;
        }
#if PT_SHADE_DYNAMIC
        if (!flush) chunk = (uint32_t)__builtin_amdgcn_readfirstlane((int)next_chunk);
#endif
    }
    flush_counters(counters, threadIdx.x & 63, 0, n_bounce, n_shadow, n_hits, st);
}

// occlusion traversal of the shadow queue (TraceShadowRay :724-742); writes the transmission next to its pending term.
template <bool COUNT>
__global__ __launch_bounds__(kBlock, PT_TRACE_WAVES) void k_wf_shadow(SceneRec sc, WfBuffers wf, int bounce, uint32_t flags, float tmax, Counters* __restrict__ counters) {
    __shared__ int s_stack[kStackLds * kBlock];
    stage_luts(sc);
    const ShardView sv = shard_view(wf);
    LaneStats st = {0, 0, 0, 0};
    trace_persistent<COUNT, 1>(sc, wf, s_stack + threadIdx.x, sv, shadow_counter(bounce), 0, 0xff, flags, st, tmax, wf.occ_cache != nullptr && bounce == 0);
    if (COUNT) { flush_counters(counters, threadIdx.x & 63, 0, 0, 0, 0, st, true); if (st.deep) atomicAdd(&counters->deep_pushes, (unsigned long long)st.deep); }
    else if (st.overflow | st.deep) flush_rare(counters, st);
}

__global__ __launch_bounds__(kBlock) void k_wf_resolve(FrameConstants fc, WfBuffers wf, float4* __restrict__ output) {
    const uint32_t pslot = blockIdx.x * kBlock + threadIdx.x;       // pixel slot; its samples sit pixel_slots apart
    uint32_t px, py;
    if (pslot >= fc.pixel_slots || !slot_pixel(fc, pslot, px, py)) return;
    // the samples of a batch are blended in sample order, exactly as consecutive PathtraceScene calls would (running mean); the
    // pixel stays in registers between them: one read and one write of the image however many samples the batch has
    float4* outp = output + ((size_t)py * fc.res_x + px);
    const bool accumulate = (fc.flags & PT_FLAG_ACCUMULATE) != 0;
    float4 pixel = make_float4(0, 0, 0, 0);
    if (accumulate && fc.accumulated_frames != 0) pixel = *outp;
    for (uint32_t k = 0; k < fc.spp; k++) {
        const uint32_t slot = k * fc.pixel_slots + pslot;
        float4 Lq = SLD(wf.L[SIDX(slot)]);
        vec3 L = v3(Lq.x, Lq.y, Lq.z);
        apply_pending(wf, slot, __float_as_uint(Lq.w), L);
        L = sanitize_sample(fc, L);
        const int accumulated = fc.accumulated_frames + (int)k;
        pixel = (accumulate && accumulated != 0) ? blend_sample(pixel, accumulated, L) : make_float4(L.x, L.y, L.z, 1.0f);
    }
    *outp = pixel;
}

// ---- host side ------------------------------------------------------------------------------------------------------------
static uint32_t blocks_per_shard_for(int stage_blocks) {
    uint32_t b = (uint32_t)(stage_blocks > 0 ? stage_blocks : 1536) / kShards;
    return b < 1 ? 1 : b;
}
// k_wf_generate: every XCD covers its band of tiles x spp samples with its kShards * blocks_per_shard / 8 workgroups
static uint32_t gen_region_tiles_for(const FrameConstants& fc) {
#if PT_GEN_XCD_BANDS == 2
    if (fc.tile_rank_count == 1) return ((fc.tiles_x + kXcds - 1) / kXcds) * fc.tiles_y;      // the widest column band
#endif
    return (fc.my_tiles + kXcds - 1) / kXcds;
}
static uint32_t gen_rounds_for(const FrameConstants& fc, uint32_t blocks_per_shard) {
    const uint32_t per_xcd = kShards * blocks_per_shard / kXcds;
    const uint32_t pairs = gen_region_tiles_for(fc) * fc.spp;
    return (pairs + per_xcd - 1) / per_xcd;
}
static uint32_t seg_cap_for(const FrameConstants& fc, uint32_t blocks_per_shard) {
    // a generate round gives shard s one tile from each of its blocks_per_shard workgroups {s, s + kShards, ...}
    return gen_rounds_for(fc, blocks_per_shard) * blocks_per_shard * kBlock;
}

// entries of each path-state array: whole 256-slot chunks, the same number for every shard (state_index)
static uint32_t chunks_per_shard_for(size_t slots) { return (uint32_t)(((slots + kBlock - 1) / kBlock + kShards - 1) / kShards); }
static size_t state_slots_for(size_t slots) { return (size_t)chunks_per_shard_for(slots) * kShards * kBlock; }

size_t wavefront_workspace_bytes(const FrameConstants& fc, int stage_blocks) {
    const uint32_t bps = blocks_per_shard_for(stage_blocks);
    const size_t slots = state_slots_for((size_t)fc.my_tiles * kBlock * fc.spp);
    const size_t q = (size_t)kShards * seg_cap_for(fc, bps);
    return slots * (5 * 16) + q * (4 * 16 + 4 * 16 + 16 + 2 * 16 + 2 * 2 * 16 + (PT_OCC_CACHE ? 2 * 4 : 0)) + kCounterArrays * kShards * kCounterStride * 4 + 52 * 256;
}

static WfBuffers carve(void* base, const FrameConstants& fc, int stage_blocks) {
    const uint32_t slots = fc.my_tiles * kBlock * fc.spp;
    const size_t state_slots = state_slots_for(slots);
    WfBuffers wf;
    wf.chunks_per_shard = chunks_per_shard_for(slots);
    char* p = (char*)base;
    auto take = [&](size_t bytes) { char* r = p; p += (bytes + 255) & ~(size_t)255; return r; };
    wf.blocks_per_shard = blocks_per_shard_for(stage_blocks);
    wf.seg_cap = seg_cap_for(fc, wf.blocks_per_shard);
    wf.gen_region_tiles = gen_region_tiles_for(fc);
    wf.gen_rounds = gen_rounds_for(fc, wf.blocks_per_shard);
    const size_t q = (size_t)kShards * wf.seg_cap;
    for (int k = 0; k < kCounterArrays; k++) wf.cnt[k] = (uint32_t*)take((size_t)kShards * kCounterStride * 4);
    wf.L = (float4*)take(state_slots * 16);
    wf.beta_pdf = (float4*)take(state_slots * 16);
    wf.thr_misc = (float4*)take(state_slots * 16);
    wf.pend = (float4*)take(state_slots * 32);
    for (int k = 0; k < 2; k++) { wf.ray_o[k] = (float4*)take(q * 16); wf.ray_d[k] = (float4*)take(q * 16); }
    for (int k = 0; k < 2; k++) { wf.q_beta[k] = (float4*)take(q * 16); wf.q_thr[k] = (float4*)take(q * 16); }
    wf.hit = (float4*)take(q * 16);
    wf.env_a = (float4*)take(q * 16);
    wf.env_b = (float4*)take(q * 16);
    wf.sh_o = (float4*)take(q * 2 * 16);
    wf.sh_d = (float4*)take(q * 2 * 16);
    wf.sh_c = PT_OCC_CACHE ? (uint32_t*)take(q * 2 * 4) : nullptr;
    wf.occ_cache = nullptr;
    wf.capacity = slots;
    return wf;
}

int traversal_stack_capacity() { return kStackLds + kStackSpill; }
size_t traversal_grid_lanes(int stage_blocks) { return (size_t)kShards * blocks_per_shard_for(stage_blocks) * kBlock; }

hipError_t launch_wavefront(const SceneRec& sc, const FrameConstants& fc, float4* output, Counters* counters, bool count, void* workspace,
                            int stage_blocks, StageTimers* timers, hipStream_t stream, uint32_t* occ_cache) {
    if (timers) timers->used = 0;
    if (fc.my_tiles == 0) return hipSuccess;
    // pt_enable_stage_timing: an event after every launch, so that the time of a launch can be split by stage (diagnostic: the
    // events cost a few microseconds each)
    auto event_at = [&](size_t i) -> hipEvent_t {
        while (timers->ev.size() <= i) { hipEvent_t e = nullptr; if (hipEventCreate(&e) != hipSuccess) return nullptr; timers->ev.push_back(e); }
        return timers->ev[i];
    };
    auto mark = [&](int kind) {                              // the launch just enqueued was of stage `kind`
        if (!timers) return;
        const size_t k = timers->used;
        hipEvent_t ev = event_at(k + 1);
        if (!ev) return;
        if (timers->kind.size() < k + 1) timers->kind.resize(k + 1);
        timers->kind[k] = (uint8_t)kind;
        hipEventRecord(ev, stream);
        timers->used = k + 1;
    };
    const uint32_t slots = fc.my_tiles * kBlock * fc.spp;
    WfBuffers wf = carve(workspace, fc, stage_blocks);
    wf.occ_cache = PT_OCC_CACHE ? occ_cache : nullptr;
    hipError_t e = hipMemsetAsync(wf.cnt[0], 0, (size_t)kCounterArrays * kShards * kCounterStride * 4, stream);     // the counter arrays are contiguous
    if (e) return e;
    const dim3 block(kBlock), full(fc.my_tiles), stage(kShards * wf.blocks_per_shard);
    if (timers) { hipEvent_t ev = event_at(0); if (ev) hipEventRecord(ev, stream); else timers = nullptr; }
    hipLaunchKernelGGL(k_wf_generate, stage, block, 0, stream, fc, wf, counters);
    mark(STAGE_GENERATE);
    const uint32_t flags = fc.flags;
    const int iterations = fc.debug_output != PT_DEBUG_OUTPUT_NONE ? 1 : fc.max_bounces + 1;
    auto ray_flags = [&](int b, uint32_t& rf, uint32_t& rmask) {
        rmask = 0xff;
        if (b == 0) rf = (flags & PT_FLAG_CULL_BACKFACE) ? RF_CULL_BACK : 0;                          // RayGeneration :747
        else {                                                                                        // TraceBounceRay :671-672
            rf = (flags & PT_FLAG_CULL_BACKFACE) ? RF_CULL_FRONT : 0;
            rmask = (flags & PT_FLAG_INDIRECT_ENVIRONMENT_ONLY) ? 0 : 0xff;
        }
    };
    // the traversal kernels compiled for rays with no flags, if that is what this frame's settings give them
    const bool trav_defaults = PT_TRAV_SPECIALISE && (flags & kTravFlagMask) == 0;
    auto launch_trace = [&](dim3 grid, const WfBuffers& w, int cur, int b, uint32_t rf, uint32_t rmask) {
        if (trav_defaults) { if (count) hipLaunchKernelGGL((k_wf_trace<true, true>), grid, block, 0, stream, sc, fc, w, cur, b, rf, rmask, counters);
                             else hipLaunchKernelGGL((k_wf_trace<false, true>), grid, block, 0, stream, sc, fc, w, cur, b, rf, rmask, counters); }
        else { if (count) hipLaunchKernelGGL((k_wf_trace<true, false>), grid, block, 0, stream, sc, fc, w, cur, b, rf, rmask, counters);
               else hipLaunchKernelGGL((k_wf_trace<false, false>), grid, block, 0, stream, sc, fc, w, cur, b, rf, rmask, counters); }
    };
    auto launch_traverse = [&](dim3 grid, const WfBuffers& w, int nxt, int b, uint32_t rf, uint32_t rmask) {
        if (trav_defaults) { if (count) hipLaunchKernelGGL((k_wf_traverse<true, true>), grid, block, 0, stream, sc, fc, w, nxt, b, rf, rmask, flags, counters);
                             else hipLaunchKernelGGL((k_wf_traverse<false, true>), grid, block, 0, stream, sc, fc, w, nxt, b, rf, rmask, flags, counters); }
        else { if (count) hipLaunchKernelGGL((k_wf_traverse<true, false>), grid, block, 0, stream, sc, fc, w, nxt, b, rf, rmask, flags, counters);
               else hipLaunchKernelGGL((k_wf_traverse<false, false>), grid, block, 0, stream, sc, fc, w, nxt, b, rf, rmask, flags, counters); }
    };
    // the shade kernel compiled for this frame's flags, if there is one (k_wf_shade)
    const uint32_t shade_bits = flags & kShadeFlagMask;
    const int shade_variant = (!PT_SHADE_SPECIALISE || fc.debug_output != PT_DEBUG_OUTPUT_NONE) ? 0 : (shade_bits == kShadeDefaults ? 1 : (shade_bits == kShadeDefaultsNoLights ? 2 : 0));
    const bool small_tables = PT_SHADE_SPECIALISE && sc.n_instances <= kInstCacheMax && sc.n_materials <= kMatCacheMax && fc.num_of_lights <= kLightCacheMax;
    auto launch_shade = [&](dim3 grid, const WfBuffers& w, int cur, int b) {
        if (small_tables && shade_variant == 1) hipLaunchKernelGGL((k_wf_shade<kShadeSpecialised | kShadeDefaults, true>), grid, block, 0, stream, sc, fc, w, cur, b, counters);
        else if (small_tables && shade_variant == 2) hipLaunchKernelGGL((k_wf_shade<kShadeSpecialised | kShadeDefaultsNoLights, true>), grid, block, 0, stream, sc, fc, w, cur, b, counters);
        else if (small_tables) hipLaunchKernelGGL((k_wf_shade<0u, true>), grid, block, 0, stream, sc, fc, w, cur, b, counters);
        else hipLaunchKernelGGL((k_wf_shade<0u, false>), grid, block, 0, stream, sc, fc, w, cur, b, counters);
    };
#if PT_LATE_GRID
    // Late bounces carry few paths (Russian roulette starts after min_bounces and the reference's throughput drives the continuation
    // probability to its floor): a launch sized for the full queue then mostly starts workgroups that find nothing and, in the shade
    // stage, would stage 67 KB of tables for it.  The grid of a bounce follows the EXPECTED queue (a quarter of the paths per bounce
    // beyond min_bounces + 1); only speed depends on the guess -- any multiple of kShards workgroups walks the whole queue.
    // tuning aid (tools/overlap_probe.py): MIPT_SHADE_BPS / MIPT_TRACE_BPS = workgroups per shard of the shade / traversal launches
    static const int env_shade_bps = getenv("MIPT_SHADE_BPS") ? atoi(getenv("MIPT_SHADE_BPS")) : 0;
    static const int env_trace_bps = getenv("MIPT_TRACE_BPS") ? atoi(getenv("MIPT_TRACE_BPS")) : 0;
    auto grid_of = [&](int b) -> dim3 {
        uint32_t bps = wf.blocks_per_shard;
        if (b > fc.min_bounces + 1) {
            double expected = (double)slots;
            for (int k = fc.min_bounces + 1; k < b; k++) expected *= 0.25;
            const uint32_t want = expected >= 1200000.0 ? 6u : (expected >= 400000.0 ? 3u : 2u);
            bps = want < bps ? want : bps;
        }
        return dim3(kShards * bps);
    };
#define PT_GRID(b) grid_of(b)
#define PT_WF(b) wf_for(b)
    auto wf_for = [&](int b) { WfBuffers w = wf; w.blocks_per_shard = grid_of(b).x / kShards; return w; };
    auto cap = [&](dim3 g, int env) { if (env > 0 && (uint32_t)env * kShards < g.x) g.x = (uint32_t)env * kShards; return g; };
    auto wf_of = [&](dim3 g) { WfBuffers w = wf; w.blocks_per_shard = g.x / kShards; return w; };
#else
#define PT_GRID(b) stage
#define PT_WF(b) wf
#endif
#if PT_FUSE_TRAVERSAL
    // generate -> trace(0) -> [shade(b) -> shadow(b) + trace(b + 1)] x (bounces) -> resolve: two grid-wide synchronisations per bounce
    {
        uint32_t rf, rmask;
        ray_flags(0, rf, rmask);
#if PT_LATE_GRID
        const dim3 g0 = cap(stage, env_trace_bps);
        launch_trace(g0, wf_of(g0), 0, 0, rf, rmask);
#else
        launch_trace(stage, wf, 0, 0, rf, rmask);
#endif
        mark(STAGE_TRACE);
    }
    for (int b = 0; b < iterations; b++) {
        const int cur = b & 1;
#if PT_LATE_GRID
        const dim3 gs = cap(grid_of(b), env_shade_bps), gt = cap(grid_of(b), env_trace_bps);
        const WfBuffers ws = wf_of(gs), wt = wf_of(gt);
#else
        const dim3 gs = stage, gt = stage;
        const WfBuffers& ws = wf; const WfBuffers& wt = wf;
#endif
        launch_shade(gs, ws, cur, b);
        mark(STAGE_SHADE);
        uint32_t rf, rmask;
        ray_flags(b + 1, rf, rmask);
        if (b + 1 < iterations) {
            launch_traverse(gt, wt, cur ^ 1, b, rf, rmask);
        } else {                                                                                      // the last vertex pushes no bounce ray
            if (count) hipLaunchKernelGGL(k_wf_shadow<true>, gt, block, 0, stream, sc, wt, b, flags, fc.max_ray_length, counters);
            else hipLaunchKernelGGL(k_wf_shadow<false>, gt, block, 0, stream, sc, wt, b, flags, fc.max_ray_length, counters);
        }
        mark(STAGE_SHADOW);
    }
#else
    for (int b = 0; b < iterations; b++) {
        const int cur = b & 1;
        uint32_t rf, rmask;
        ray_flags(b, rf, rmask);
        launch_trace(stage, wf, cur, b, rf, rmask);
        mark(STAGE_TRACE);
        launch_shade(stage, wf, cur, b);
        mark(STAGE_SHADE);
        if (count) hipLaunchKernelGGL(k_wf_shadow<true>, stage, block, 0, stream, sc, wf, b, flags, fc.max_ray_length, counters);
        else hipLaunchKernelGGL(k_wf_shadow<false>, stage, block, 0, stream, sc, wf, b, flags, fc.max_ray_length, counters);
        mark(STAGE_SHADOW);
    }
#endif
    hipLaunchKernelGGL(k_wf_resolve, full, block, 0, stream, fc, wf, output);
    mark(STAGE_RESOLVE);
    return hipGetLastError();
}

}  // namespace pt
