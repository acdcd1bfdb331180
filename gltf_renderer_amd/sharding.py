"""Pixel-tile sharding of one frame across the GPUs of a node (new capability).

The reference is single-GPU.  Every pixel-sample is independent (its RNG stream is keyed by
(pixel, seed, counter), PathTracer.lib.hlsl:146), so a frame shards with no data-path exchange:
16x16 tiles in row-major order, tile t belongs to rank t % N (interleaved for load balance), each rank
renders its tiles into a full-size RGBA32F image, and ONE exchange per frame assembles it on rank 0.

Two equivalent forms of that exchange (tiles are disjoint, so both are bit-identical to a 1-rank frame):
  reduce_frame  - reduce(sum) of the zeroed full-size image: one RCCL ring/tree reduce of W*H*16 B
  gather_frame  - each rank packs only ITS tiles (1/N of the image) and rank 0 receives them point to
                  point.  xGMI is a full mesh of direct links, so the N-1 senders use N-1 different links
                  in parallel and each moves 1/N of the bytes: ~(N-1)/N of the image crosses the fabric
                  once, instead of a ring pushing the whole image through every link.
The same functions run over RCCL on GPUs (backend "nccl") and over gloo in the CPU tests.
"""
import torch
import torch.distributed as dist

TILE = 16


def tile_owner(tile, world):
    return tile % world


def my_tile_count(width, height, rank, world):
    tiles = ((width + TILE - 1) // TILE) * ((height + TILE - 1) // TILE)
    return (tiles - rank + world - 1) // world if tiles > rank else 0


def reduce_frame(image, world, dst=0):
    """image: torch tensor (H, W, 4) holding this rank's tiles (zeros elsewhere)."""
    if world > 1:
        dist.reduce(image, dst=dst, op=dist.ReduceOp.SUM)
    return image


class TileExchange:
    """Pixel index lists of every rank's tiles for one (width, height, world); built once, reused every frame."""

    def __init__(self, width, height, world, device):
        self.width, self.height, self.world = width, height, world
        tx = (width + TILE - 1) // TILE
        y, x = torch.meshgrid(torch.arange(height), torch.arange(width), indexing="ij")
        owner = ((y // TILE) * tx + (x // TILE)) % world
        flat = torch.arange(height * width).reshape(height, width)
        self.index = [flat[owner == r].to(device) for r in range(world)]
        self.count = [int(i.numel()) for i in self.index]
        self.max_count = max(self.count)

    def gather_frame(self, image, rank, dst=0):
        """image: (H, W, 4) float32 with this rank's tiles rendered (other pixels: anything).  After the call rank `dst`
        holds the assembled frame."""
        if self.world == 1:
            return image
        px = image.view(-1, 4)
        send = torch.zeros((self.max_count, 4), dtype=image.dtype, device=image.device)
        send[: self.count[rank]] = px.index_select(0, self.index[rank])
        if rank == dst:
            parts = [torch.empty_like(send) for _ in range(self.world)]
            dist.gather(send, parts, dst=dst)
            for r in range(self.world):
                if r != dst:
                    px.index_copy_(0, self.index[r], parts[r][: self.count[r]])
        else:
            dist.gather(send, None, dst=dst)
        return image
