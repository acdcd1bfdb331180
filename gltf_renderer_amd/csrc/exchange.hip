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
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
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

Rccl* rccl() {
    static Rccl r;
    if (r.lib || !r.error.empty()) return &r;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (r.lib) break;
    }
    if (!r.lib) { r.error = std::string("cannot load librccl.so.1: ") + (dlerror() ? dlerror() : "not found"); return &r; }
#define PT_SYM(field, name)                                                                    \
    r.field = (decltype(r.field))dlsym(r.lib, name);                                           \
    if (!r.field && r.error.empty()) r.error = std::string("librccl lacks ") + name;
    PT_SYM(GetUniqueId, "ncclGetUniqueId") PT_SYM(CommInitRank, "ncclCommInitRank") PT_SYM(CommDestroy, "ncclCommDestroy")
    PT_SYM(GroupStart, "ncclGroupStart") PT_SYM(GroupEnd, "ncclGroupEnd") PT_SYM(Send, "ncclSend") PT_SYM(Recv, "ncclRecv")
    PT_SYM(Reduce, "ncclReduce") PT_SYM(GetErrorString, "ncclGetErrorString")
#undef PT_SYM
    if (!r.error.empty()) { dlclose(r.lib); r.lib = nullptr; }
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

struct ExchangeState {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    void* send = nullptr; size_t send_cap = 0;          // this rank's packed tiles / the zero-masked copy
    void* recv = nullptr; size_t recv_cap = 0;          // root: the other ranks' packed tiles, back to back
};

void exchange_free(ExchangeState* x) {
    if (!x) return;
    if (x->comm && rccl()->CommDestroy) rccl()->CommDestroy(x->comm);
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

int exchange_create(ExchangeState** out, int rank, int world, const void* id128, std::string& err) {
    *out = nullptr;
    ExchangeState* x = new ExchangeState();
    x->rank = rank; x->world = world;
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
    const bool root = rank == dst;
    auto hipfail = [&](const char* what, hipError_t e) { err = std::string(what) + ": " + hipGetErrorString(e); return PT_ERR_DEVICE; };
    auto ncclfail = [&](const char* what, ncclResult_t q) { err = std::string(what) + ": " + rccl()->GetErrorString(q); return PT_ERR_DEVICE; };
    hipError_t e;
    if (world == 1) {                                    // nothing to exchange: the local image is the frame
        if (frame && frame != local && (e = hipMemcpyAsync(frame, local, (size_t)w * h * 16, hipMemcpyDeviceToDevice, stream))) return hipfail("hipMemcpyAsync", e);
        if (!x->comm) return PT_OK;
    }
    Rccl* r = rccl();
    if (mode == PT_EXCHANGE_REDUCE) {
        const size_t bytes = (size_t)w * h * 16;
        if ((e = grow(x->send, x->send_cap, bytes, stream))) return hipfail("hipMalloc", e);
        hipLaunchKernelGGL(k_tiles_mask, dim3((w + PT_TILE - 1) / PT_TILE, (h + PT_TILE - 1) / PT_TILE), dim3(256), 0, stream, (const float4*)local, w, h,
                           (uint32_t)rank, (uint32_t)world, (w + PT_TILE - 1) / PT_TILE, (float4*)x->send);
        if ((e = hipGetLastError())) return hipfail("k_tiles_mask", e);
        ncclResult_t q = r->Reduce(x->send, root ? frame : x->send, (size_t)w * h * 4, ncclFloat, ncclSum, dst, x->comm, stream);
        if (q != ncclSuccess) return ncclfail("ncclReduce", q);
        return PT_OK;
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
    ncclResult_t q = r->GroupStart();
    if (q != ncclSuccess) return ncclfail("ncclGroupStart", q);
    if (world == 1) {                                    // a communicator of one: send to self, to run the calls (1-GPU boxes)
        if (mine) { q = r->Send(x->send, mine / 4, ncclFloat, 0, x->comm, stream); if (q == ncclSuccess) q = r->Recv(x->recv, mine / 4, ncclFloat, 0, x->comm, stream); }
    } else if (root) {
        for (int k = 0; k < world && q == ncclSuccess; k++)
            if (k != rank && offset[k + 1] > offset[k]) q = r->Recv((char*)x->recv + offset[k], (offset[k + 1] - offset[k]) / 4, ncclFloat, k, x->comm, stream);
    } else if (mine) q = r->Send(x->send, mine / 4, ncclFloat, dst, x->comm, stream);
    ncclResult_t q2 = r->GroupEnd();
    if (q != ncclSuccess) return ncclfail("ncclSend/ncclRecv", q);
    if (q2 != ncclSuccess) return ncclfail("ncclGroupEnd", q2);
    if (root)
        for (int k = 0; k < world; k++)
            if ((k != rank || world == 1) && offset[k + 1] > offset[k] && (e = tiles_unpack((char*)x->recv + offset[k], w, h, k, world, frame, stream)))
                return hipfail("k_tiles_unpack", e);
    return PT_OK;
}

}  // namespace pt
