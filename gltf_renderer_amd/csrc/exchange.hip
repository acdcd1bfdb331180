// exchange.hip -- the ONE exchange per output frame of the tile-sharded renderer, in the C/C++ host.
//
// The reference is single-GPU (Source/Renderer.cpp:56).  Here one process per GPU renders the 16x16 tiles t with
// t % world == rank into its own full-size accumulation image; per reported frame the tiles are assembled on one rank by a
// single RCCL exchange on the context's stream:
//
//   PT_EXCHANGE_GATHER  every rank packs only ITS tiles (1/N of the image, slot order) into a persistent buffer; the root posts
//                       N-1 receives, every other rank one send, in one RCCL group.  xGMI is a full mesh of point-to-point links,
//                       so the N-1 senders use N-1 different links in parallel and ~(N-1)/N of the image crosses the fabric once.
//   PT_EXCHANGE_REDUCE  the form north_star names: ncclReduce(sum) of a ZERO-MASKED COPY of the image (own tiles, zeros
//                       elsewhere) into the root's frame -- never of the accumulation target itself, which would double-count the
//                       other ranks' tiles on the root once frames accumulate.
//
// Either way each rank's accumulation image stays private, so FLAG_ACCUMULATE composes with the exchange over any number of frames.
// RCCL is bound at run time (dlopen of librccl.so.1: the copy PyTorch has already mapped when the host is Python, the ROCm one
// otherwise), so libmipt.so has no link-time dependency on it and a single-GPU host never loads it.
//
// exchange_frame talks to its peers through a small TRANSPORT table (prepare / begin / send / recv / end / reduce):
//   rccl      one process per GPU, the production transport: grouped ncclSend / ncclRecv, ncclReduce;
//   loopback  N contexts of ONE process (on one GPU or on several) joined by pt_exchange_create_loopback: a send posts its buffer
//             and an event, the matching receive is a stream-ordered device-to-device copy on the receiver's stream, the reduce
//             sums the posted buffers in rank order.  It is how a single-process multi-context host exchanges, and how the rank
//             logic of exchange_frame (per-rank offsets, ragged tile counts, dst != 0, accumulation over frames) is tested for
//             N = 2, 3, 8 on a one-GPU box.  Posting is not blocking, so the root must be called AFTER the other ranks.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "pt_host.h"

namespace pt {

namespace {

struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclReduce) Reduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};

// MIPT_RCCL_LIBRARY names the one library to try instead of the default search list (a private RCCL build; the host-ABI test points it
// at a missing file to exercise the PT_ERR_NOT_READY path).
void rccl_load(Rccl& r) {
    const char* override_name = getenv("MIPT_RCCL_LIBRARY");
    const char* defaults[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    std::string tried;
    for (const char* n : defaults) {
        if (override_name) n = override_name;
        r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (r.lib) break;
        const char* m = dlerror();                       // one call: dlerror() clears the message it returns
        tried = std::string("cannot load ") + n + ": " + (m ? m : "not found");
        if (override_name) break;
    }
    if (!r.lib) { r.error = tried; return; }
#define PT_SYM(field, name)                                                                    \
    r.field = (decltype(r.field))dlsym(r.lib, name);                                           \
    if (!r.field && r.error.empty()) r.error = std::string("librccl lacks ") + name;
    PT_SYM(GetUniqueId, "ncclGetUniqueId") PT_SYM(CommInitRank, "ncclCommInitRank") PT_SYM(CommDestroy, "ncclCommDestroy")
    PT_SYM(GroupStart, "ncclGroupStart") PT_SYM(GroupEnd, "ncclGroupEnd") PT_SYM(Send, "ncclSend") PT_SYM(Recv, "ncclRecv")
    PT_SYM(Reduce, "ncclReduce") PT_SYM(GetErrorString, "ncclGetErrorString")
#undef PT_SYM
    if (!r.error.empty()) { dlclose(r.lib); r.lib = nullptr; }
}

Rccl* rccl() {                                           // rank threads of one process may race here: filled exactly once
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] { rccl_load(r); });
    return &r;
}

// Tiles of rank r in slot order: local tile k is tile r + k * world (row-major over the tile grid), 256 slots per tile, one wave
// per 8x8 quadrant -- the renderer's own slot -> pixel map (pt_vertex.h slot_pixel), so a packed buffer is the image in the order
// the kernels produced it.
__device__ __forceinline__ bool tile_slot_pixel(uint32_t slot, uint32_t rank, uint32_t world, uint32_t tiles_x, uint32_t tiles_y, uint32_t w, uint32_t h,
                                                uint32_t& px, uint32_t& py) {
    const uint32_t tile = rank + (slot >> 8) * world, t = slot & 255u;
    if (tile >= tiles_x * tiles_y) return false;
    const uint32_t wave = t >> 6, lane = t & 63u;
    px = (tile % tiles_x) * PT_TILE + (wave & 1u) * 8u + (lane & 7u);
    py = (tile / tiles_x) * PT_TILE + (wave >> 1) * 8u + (lane >> 3);
    return px < w && py < h;
}

// image -> packed (pixels outside the image, in ragged edge tiles, pack as zeros)
__global__ __launch_bounds__(256) void k_tiles_pack(const float4* __restrict__ image, uint32_t w, uint32_t h, uint32_t rank, uint32_t world,
                                                    uint32_t tiles_x, uint32_t tiles_y, float4* __restrict__ packed) {
    const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    uint32_t px, py;
    packed[slot] = tile_slot_pixel(slot, rank, world, tiles_x, tiles_y, w, h, px, py) ? image[(size_t)py * w + px] : make_float4(0, 0, 0, 0);
}
// packed -> image (only that rank's pixels are written)
__global__ __launch_bounds__(256) void k_tiles_unpack(const float4* __restrict__ packed, uint32_t w, uint32_t h, uint32_t rank, uint32_t world,
                                                      uint32_t tiles_x, uint32_t tiles_y, float4* __restrict__ image) {
    const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    uint32_t px, py;
    if (tile_slot_pixel(slot, rank, world, tiles_x, tiles_y, w, h, px, py)) image[(size_t)py * w + px] = packed[slot];
}
// zero-masked copy for the reduce: own tiles from the image, zeros elsewhere
__global__ __launch_bounds__(256) void k_tiles_mask(const float4* __restrict__ image, uint32_t w, uint32_t h, uint32_t rank, uint32_t world, uint32_t tiles_x,
                                                    float4* __restrict__ out) {
    const uint32_t x = blockIdx.x * 16u + (threadIdx.x & 15u), y = blockIdx.y * 16u + (threadIdx.x >> 4);
    if (x >= w || y >= h) return;
    const uint32_t tile = blockIdx.y * tiles_x + blockIdx.x;
    const size_t i = (size_t)y * w + x;
    out[i] = (tile % world == rank) ? image[i] : make_float4(0, 0, 0, 0);
}

}  // namespace

uint32_t tiles_of_rank(uint32_t w, uint32_t h, uint32_t rank, uint32_t world) {
    const uint32_t n = ((w + PT_TILE - 1) / PT_TILE) * ((h + PT_TILE - 1) / PT_TILE);
    return n > rank ? (n - rank + world - 1) / world : 0;
}

hipError_t tiles_pack(const void* image, uint32_t w, uint32_t h, uint32_t rank, uint32_t world, void* packed, hipStream_t stream) {
    const uint32_t n = tiles_of_rank(w, h, rank, world);
    if (n) hipLaunchKernelGGL(k_tiles_pack, dim3(n), dim3(256), 0, stream, (const float4*)image, w, h, rank, world, (w + PT_TILE - 1) / PT_TILE, (h + PT_TILE - 1) / PT_TILE, (float4*)packed);
    return hipGetLastError();
}
hipError_t tiles_unpack(const void* packed, uint32_t w, uint32_t h, uint32_t rank, uint32_t world, void* image, hipStream_t stream) {
    const uint32_t n = tiles_of_rank(w, h, rank, world);
    if (n) hipLaunchKernelGGL(k_tiles_unpack, dim3(n), dim3(256), 0, stream, (const float4*)packed, w, h, rank, world, (w + PT_TILE - 1) / PT_TILE, (h + PT_TILE - 1) / PT_TILE, (float4*)image);
    return hipGetLastError();
}

// ---- transports ---------------------------------------------------------------------------------------------------------------------
// loopback transport: what the contexts of one group share
struct LoopPost {
    const void* buf = nullptr; size_t n = 0; hipEvent_t ready = nullptr; bool valid = false;
    hipEvent_t consumed = nullptr; bool consumed_pending = false; int consumer_device = 0;     // recorded by the receiver after its copy
};
struct LoopWorld {
    std::mutex m;
    int world = 0, members = 0;
    std::vector<LoopPost> p2p;                          // [src * world + dst]
    std::vector<LoopPost> red;                          // [src]
    ~LoopWorld() { for (auto* v : {&p2p, &red}) for (auto& p : *v) if (p.consumed) hipEventDestroy(p.consumed); }
};
struct ExchangeState {
    const struct Transport* via = nullptr;
    ncclComm_t comm = nullptr;                          // rccl transport
    std::shared_ptr<LoopWorld> loop;                    // loopback transport
    hipEvent_t posted = nullptr;                        // loopback: "my send buffer is ready", recorded on my stream
    int device = 0;
    int rank = 0, world = 1;
    void* send = nullptr; size_t send_cap = 0;          // this rank's packed tiles / the zero-masked copy
    void* recv = nullptr; size_t recv_cap = 0;          // root: the other ranks' packed tiles, back to back
};

// What exchange_frame asks of a transport.  Counts are in floats; every call is asynchronous on `s`.
struct Transport {
    const char* name;
    int (*prepare)(ExchangeState& x, hipStream_t s, std::string& err);      // before the send buffer is rewritten
    int (*begin)(ExchangeState& x, std::string& err);
    int (*send)(ExchangeState& x, const void* buf, size_t n, int peer, hipStream_t s, std::string& err);
    int (*recv)(ExchangeState& x, void* buf, size_t n, int peer, hipStream_t s, std::string& err);
    int (*end)(ExchangeState& x, hipStream_t s, std::string& err);
    int (*reduce)(ExchangeState& x, const void* sendbuf, void* recvbuf, size_t n, int root, hipStream_t s, std::string& err);
};

namespace {

// -- rccl
int nccl_rc(const char* what, ncclResult_t q, std::string& err) {
    if (q == ncclSuccess) return PT_OK;
    err = std::string(what) + ": " + rccl()->GetErrorString(q);
    return PT_ERR_DEVICE;
}
const Transport k_rccl = {
    "rccl",
    [](ExchangeState&, hipStream_t, std::string&) { return (int)PT_OK; },
    [](ExchangeState&, std::string& err) { return nccl_rc("ncclGroupStart", rccl()->GroupStart(), err); },
    [](ExchangeState& x, const void* buf, size_t n, int peer, hipStream_t s, std::string& err) { return nccl_rc("ncclSend", rccl()->Send(buf, n, ncclFloat, peer, x.comm, s), err); },
    [](ExchangeState& x, void* buf, size_t n, int peer, hipStream_t s, std::string& err) { return nccl_rc("ncclRecv", rccl()->Recv(buf, n, ncclFloat, peer, x.comm, s), err); },
    [](ExchangeState&, hipStream_t, std::string& err) { return nccl_rc("ncclGroupEnd", rccl()->GroupEnd(), err); },
    [](ExchangeState& x, const void* sendbuf, void* recvbuf, size_t n, int root, hipStream_t s, std::string& err) {
        return nccl_rc("ncclReduce", rccl()->Reduce(sendbuf, recvbuf, n, ncclFloat, ncclSum, root, x.comm, s), err);
    },
};

// -- loopback: the contexts of one process that joined the same group
std::mutex g_loop_mutex;
std::map<uint64_t, std::weak_ptr<LoopWorld>> g_loop_worlds;

int hip_rc(const char* what, hipError_t e, std::string& err) {
    if (e == hipSuccess) return PT_OK;
    err = std::string(what) + ": " + hipGetErrorString(e);
    return PT_ERR_DEVICE;
}
__global__ __launch_bounds__(256) void k_add_into(float4* __restrict__ acc, const float4* __restrict__ x, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i < n4) { float4 a = acc[i], b = x[i]; acc[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
}
// the receiver took a posted buffer: remember when its copy is done, so the poster does not rewrite the buffer under it
int loop_consume(LoopPost& p, int device, hipStream_t s, std::string& err) {
    if (!p.consumed) {
        if (int rc = hip_rc("hipEventCreate", hipEventCreateWithFlags(&p.consumed, hipEventDisableTiming), err)) return rc;
        p.consumer_device = device;
    }
    if (int rc = hip_rc("hipEventRecord", hipEventRecord(p.consumed, s), err)) return rc;
    p.consumed_pending = true; p.valid = false;
    return PT_OK;
}
int loop_post(ExchangeState& x, LoopPost& p, const void* buf, size_t n, hipStream_t s, std::string& err) {
    if (p.valid) { err = "loopback transport: the previous buffer of this rank was never received (call pt_exchange_frame on the root after the other ranks, every frame)"; return PT_ERR_NOT_READY; }
    if (int rc = hip_rc("hipEventRecord", hipEventRecord(x.posted, s), err)) return rc;
    p.buf = buf; p.n = n; p.ready = x.posted; p.valid = true;
    return PT_OK;
}
const Transport k_loopback = {
    "loopback",
    [](ExchangeState& x, hipStream_t s, std::string& err) {           // my earlier posts: wait (on the stream) until their receivers are done with them
        std::lock_guard<std::mutex> g(x.loop->m);
        for (int k = 0; k < x.world; k++) {
            LoopPost& p = x.loop->p2p[(size_t)x.rank * x.world + k];
            if (p.consumed_pending) { if (int rc = hip_rc("hipStreamWaitEvent", hipStreamWaitEvent(s, p.consumed, 0), err)) return rc; p.consumed_pending = false; }
        }
        LoopPost& q = x.loop->red[x.rank];
        if (q.consumed_pending) { if (int rc = hip_rc("hipStreamWaitEvent", hipStreamWaitEvent(s, q.consumed, 0), err)) return rc; q.consumed_pending = false; }
        return (int)PT_OK;
    },
    [](ExchangeState&, std::string&) { return (int)PT_OK; },
    [](ExchangeState& x, const void* buf, size_t n, int peer, hipStream_t s, std::string& err) {
        std::lock_guard<std::mutex> g(x.loop->m);
        return loop_post(x, x.loop->p2p[(size_t)x.rank * x.world + peer], buf, n, s, err);
    },
    [](ExchangeState& x, void* buf, size_t n, int peer, hipStream_t s, std::string& err) {
        std::lock_guard<std::mutex> g(x.loop->m);
        LoopPost& p = x.loop->p2p[(size_t)peer * x.world + x.rank];
        if (!p.valid) { err = "loopback transport: rank " + std::to_string(peer) + " has not sent yet (call pt_exchange_frame on the root after the other ranks)"; return (int)PT_ERR_NOT_READY; }
        if (p.n != n) { err = "loopback transport: rank " + std::to_string(peer) + " sent " + std::to_string(p.n) + " floats, " + std::to_string(n) + " expected"; return (int)PT_ERR_INVALID_ARGUMENT; }
        if (int rc = hip_rc("hipStreamWaitEvent", hipStreamWaitEvent(s, p.ready, 0), err)) return rc;
        if (int rc = hip_rc("hipMemcpyAsync", hipMemcpyAsync(buf, p.buf, n * 4, hipMemcpyDeviceToDevice, s), err)) return rc;
        return loop_consume(p, x.device, s, err);
    },
    [](ExchangeState&, hipStream_t, std::string&) { return (int)PT_OK; },
    [](ExchangeState& x, const void* sendbuf, void* recvbuf, size_t n, int root, hipStream_t s, std::string& err) {
        std::lock_guard<std::mutex> g(x.loop->m);
        if (x.rank != root) return loop_post(x, x.loop->red[x.rank], sendbuf, n, s, err);
        for (int k = 0; k < x.world; k++)
            if (k != root && (!x.loop->red[k].valid || x.loop->red[k].n != n)) {
                err = "loopback transport: rank " + std::to_string(k) + " has not contributed to the reduce yet (call the root last)";
                return (int)PT_ERR_NOT_READY;
            }
        for (int k = 0; k < x.world; k++) {                          // rank order: sum_k buf_k (the tiles are disjoint, so any order gives the same bits)
            const void* src = k == root ? sendbuf : x.loop->red[k].buf;
            if (k != root) if (int rc = hip_rc("hipStreamWaitEvent", hipStreamWaitEvent(s, x.loop->red[k].ready, 0), err)) return rc;
            if (k == 0) { if (int rc = hip_rc("hipMemcpyAsync", hipMemcpyAsync(recvbuf, src, n * 4, hipMemcpyDeviceToDevice, s), err)) return rc; }
            else {
                hipLaunchKernelGGL(k_add_into, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, (float4*)recvbuf, (const float4*)src, n / 4);
                if (int rc = hip_rc("k_add_into", hipGetLastError(), err)) return rc;
            }
            if (k != root) if (int rc = loop_consume(x.loop->red[k], x.device, s, err)) return rc;
        }
        return (int)PT_OK;
    },
};

}  // namespace

void exchange_free(ExchangeState* x) {
    if (!x) return;
    if (x->comm && rccl()->CommDestroy) rccl()->CommDestroy(x->comm);
    if (x->loop) {
        std::lock_guard<std::mutex> g(x->loop->m);
        for (int k = 0; k < x->world; k++) x->loop->p2p[(size_t)x->rank * x->world + k].valid = false;      // my buffers go away with me
        x->loop->red[x->rank].valid = false;
        x->loop->members--;
    }
    if (x->posted) hipEventDestroy(x->posted);
    hipFree(x->send); hipFree(x->recv);
    delete x;
}

int exchange_unique_id(void* out128, std::string& err) {
    Rccl* r = rccl();
    if (!r->lib) { err = r->error; return PT_ERR_NOT_READY; }
    static_assert(sizeof(ncclUniqueId) == PT_EXCHANGE_ID_BYTES, "ncclUniqueId");
    ncclResult_t q = r->GetUniqueId((ncclUniqueId*)out128);
    if (q != ncclSuccess) { err = std::string("ncclGetUniqueId: ") + r->GetErrorString(q); return PT_ERR_DEVICE; }
    return PT_OK;
}

// PT_OK if RCCL can be loaded in this process (no communicator, no GPU call): lets every rank of a job agree BEFORE any of them enters
// the collective ncclCommInitRank.
int exchange_probe(std::string& err) {
    Rccl* r = rccl();
    if (!r->lib) { err = r->error; return PT_ERR_NOT_READY; }
    return PT_OK;
}

int exchange_create(ExchangeState** out, int rank, int world, const void* id128, std::string& err) {
    *out = nullptr;
    ExchangeState* x = new ExchangeState();
    x->rank = rank; x->world = world; x->via = &k_rccl;
    if (world > 1 || id128) {                            // a world of one needs no communicator (and no RCCL) unless the caller asks for one
        Rccl* r = rccl();
        if (!r->lib) { err = r->error; delete x; return PT_ERR_NOT_READY; }
        if (!id128) { err = "pt_exchange_create: unique id is null"; delete x; return PT_ERR_INVALID_ARGUMENT; }
        ncclUniqueId id;
        memcpy(&id, id128, sizeof(id));
        ncclResult_t q = r->CommInitRank(&x->comm, world, id, rank);
        if (q != ncclSuccess) { err = std::string("ncclCommInitRank: ") + r->GetErrorString(q); delete x; return PT_ERR_DEVICE; }
    }
    *out = x;
    return PT_OK;
}

int exchange_create_loopback(ExchangeState** out, int rank, int world, uint64_t group, int device, std::string& err) {
    *out = nullptr;
    std::shared_ptr<LoopWorld> lw;
    {
        std::lock_guard<std::mutex> g(g_loop_mutex);
        lw = g_loop_worlds[group].lock();
        if (!lw) {
            lw = std::make_shared<LoopWorld>();
            lw->world = world; lw->p2p.resize((size_t)world * world); lw->red.resize((size_t)world);
            g_loop_worlds[group] = lw;
        }
    }
    std::lock_guard<std::mutex> g(lw->m);
    if (lw->world != world) { err = "pt_exchange_create_loopback: group " + std::to_string(group) + " exists with world " + std::to_string(lw->world); return PT_ERR_INVALID_ARGUMENT; }
    ExchangeState* x = new ExchangeState();
    x->rank = rank; x->world = world; x->via = &k_loopback; x->loop = lw; x->device = device;
    if (int rc = hip_rc("hipEventCreate", hipEventCreateWithFlags(&x->posted, hipEventDisableTiming), err)) { x->loop.reset(); delete x; return rc; }
    lw->members++;
    *out = x;
    return PT_OK;
}

static hipError_t grow(void*& p, size_t& cap, size_t need, hipStream_t stream) {
    if (need <= cap) return hipSuccess;
    hipError_t e = hipStreamSynchronize(stream);
    if (e) return e;
    hipFree(p); p = nullptr; cap = 0;
    if ((e = hipMalloc(&p, need))) return e;
    cap = need;
    return hipSuccess;
}

int exchange_frame(ExchangeState* x, const void* local, void* frame, uint32_t w, uint32_t h, int mode, int dst, hipStream_t stream, std::string& err) {
    const int world = x->world, rank = x->rank;
    if (dst < 0 || dst >= world) { err = "pt_exchange_frame: dst_rank " + std::to_string(dst) + " outside the world of " + std::to_string(world); return PT_ERR_INVALID_ARGUMENT; }
    const bool root = rank == dst;
    const Transport& T = *x->via;
    const bool transported = x->comm != nullptr || x->loop != nullptr;
    auto hipfail = [&](const char* what, hipError_t e) { err = std::string(what) + ": " + hipGetErrorString(e); return PT_ERR_DEVICE; };
    hipError_t e;
    int rc;
    if (world == 1) {                                    // nothing to exchange: the local image is the frame
        if (frame && frame != local && (e = hipMemcpyAsync(frame, local, (size_t)w * h * 16, hipMemcpyDeviceToDevice, stream))) return hipfail("hipMemcpyAsync", e);
        if (!transported) return PT_OK;
    }
    if ((rc = T.prepare(*x, stream, err))) return rc;
    if (mode == PT_EXCHANGE_REDUCE) {
        const size_t bytes = (size_t)w * h * 16;
        if ((e = grow(x->send, x->send_cap, bytes, stream))) return hipfail("hipMalloc", e);
        hipLaunchKernelGGL(k_tiles_mask, dim3((w + PT_TILE - 1) / PT_TILE, (h + PT_TILE - 1) / PT_TILE), dim3(256), 0, stream, (const float4*)local, w, h,
                           (uint32_t)rank, (uint32_t)world, (w + PT_TILE - 1) / PT_TILE, (float4*)x->send);
        if ((e = hipGetLastError())) return hipfail("k_tiles_mask", e);
        return T.reduce(*x, x->send, root ? frame : x->send, (size_t)w * h * 4, dst, stream, err);
    }
    // gather: own tiles packed, one grouped set of point-to-point transfers, the root unpacks the others' tiles into the frame
    const size_t mine = (size_t)tiles_of_rank(w, h, rank, world) * 256 * 16;
    if (!root || world == 1) {
        if ((e = grow(x->send, x->send_cap, mine ? mine : 16, stream))) return hipfail("hipMalloc", e);
        if ((e = tiles_pack(local, w, h, rank, world, x->send, stream))) return hipfail("k_tiles_pack", e);
    }
    std::vector<size_t> offset((size_t)world + 1, 0);
    if (root) {
        for (int k = 0; k < world; k++) offset[k + 1] = offset[k] + (k == rank && world > 1 ? 0 : (size_t)tiles_of_rank(w, h, k, world) * 256 * 16);
        if ((e = grow(x->recv, x->recv_cap, offset[world] ? offset[world] : 16, stream))) return hipfail("hipMalloc", e);
        if (frame != local && world > 1 && (e = hipMemcpyAsync(frame, local, (size_t)w * h * 16, hipMemcpyDeviceToDevice, stream))) return hipfail("hipMemcpyAsync", e);
    }
    if ((rc = T.begin(*x, err))) return rc;
    if (world == 1) {                                    // a world of one: send to self, to run the calls (1-GPU boxes)
        if (mine) { rc = T.send(*x, x->send, mine / 4, 0, stream, err); if (rc == PT_OK) rc = T.recv(*x, x->recv, mine / 4, 0, stream, err); }
    } else if (root) {
        for (int k = 0; k < world && rc == PT_OK; k++)
            if (k != rank && offset[k + 1] > offset[k]) rc = T.recv(*x, (char*)x->recv + offset[k], (offset[k + 1] - offset[k]) / 4, k, stream, err);
    } else if (mine) rc = T.send(*x, x->send, mine / 4, dst, stream, err);
    std::string err2;
    const int rc2 = T.end(*x, stream, err2);             // a group that was started is always ended
    if (rc != PT_OK) return rc;
    if (rc2 != PT_OK) { err = err2; return rc2; }
    if (root)
        for (int k = 0; k < world; k++)
            if ((k != rank || world == 1) && offset[k + 1] > offset[k] && (e = tiles_unpack((char*)x->recv + offset[k], w, h, k, world, frame, stream)))
                return hipfail("k_tiles_unpack", e);
    return PT_OK;
}

const char* exchange_transport_name(const ExchangeState* x) { return x && x->via ? x->via->name : "none"; }

}  // namespace pt
