"""Pixel-tile sharding of one frame across ranks: the torch.distributed TEST DOUBLE of the exchange in libmipt.so.

The product's exchange is C/C++: `pt_exchange_create / pt_exchange_frame` (include/mipt.h, csrc/exchange.hip) call RCCL on the
context's stream.  RCCL needs one GPU per rank, so the N > 1 logic cannot run in the CPU suite or on a 1-GPU box; this module
restates the same two exchanges over `torch.distributed` (gloo in tests/test_dist_cpu.py and in `bench.py --backend gloo`
rehearsals) with the same contract:

  * tile t (16x16 pixels, row-major) belongs to rank t % N; every rank renders its tiles into ITS OWN full-size accumulation
    image, which the exchange only reads -- so FLAG_ACCUMULATE composes with the exchange over any number of frames;
  * gather_frame  - each rank packs only its tiles (1/N of the image), the root receives them point to point and writes them over
                    the other ranks' tiles of the output: bit-identical to a 1-rank frame;
  * reduce_frame  - reduce(sum) of a ZERO-MASKED COPY (own tiles, zeros elsewhere), never of the accumulation target itself:
                    reducing in place would add the other ranks' tiles into the root's target again on every later frame.
"""
import torch
import torch.distributed as dist

TILE = 16


def tile_owner(tile, world):
    return tile % world


def my_tile_count(width, height, rank, world):
    tiles = ((width + TILE - 1) // TILE) * ((height + TILE - 1) // TILE)
    return (tiles - rank + world - 1) // world if tiles > rank else 0


def tile_owner_map(width, height, world, device="cpu"):
    """(H, W) int64: the rank that renders each pixel."""
    tx = (width + TILE - 1) // TILE
    y, x = torch.meshgrid(torch.arange(height), torch.arange(width), indexing="ij")
    return (((y // TILE) * tx + (x // TILE)) % world).to(device)


class TileExchange:
    """Pixel index lists of every rank's tiles for one (width, height, world) and the persistent transfer buffers; built once,
    reused every frame."""

    def __init__(self, width, height, world, device):
        self.width, self.height, self.world = width, height, world
        owner = tile_owner_map(width, height, world)
        flat = torch.arange(height * width).reshape(height, width)
        self.index = [flat[owner == r].to(device) for r in range(world)]
        self.count = [int(i.numel()) for i in self.index]
        self.max_count = max(self.count)
        self.mask = [(owner == r).to(device) for r in range(world)]
        self.send = None
        self.parts = None
        self.masked = None

    def gather_frame(self, image, rank, dst=0, out=None):
        """image: (H, W, 4) float32, this rank's accumulation image (only its own tiles are read).  On rank `dst` the other ranks'
        tiles are written into `out` (default: into `image` itself, whose foreign tiles the root never renders) and `out` is
        returned; other ranks return None."""
        if self.world == 1:
            if out is not None and out is not image:
                out.copy_(image)
            return image if out is None else out
        px = image.view(-1, 4)
        if self.send is None:
            self.send = torch.zeros((self.max_count, 4), dtype=image.dtype, device=image.device)
        torch.index_select(px, 0, self.index[rank], out=self.send[: self.count[rank]])
        if rank != dst:
            dist.gather(self.send, None, dst=dst)
            return None
        if self.parts is None:
            self.parts = [torch.empty_like(self.send) for _ in range(self.world)]
        dist.gather(self.send, self.parts, dst=dst)
        if out is None:
            out = image
        elif out is not image:
            out.copy_(image)
        opx = out.view(-1, 4)
        for r in range(self.world):
            if r != dst:
                opx.index_copy_(0, self.index[r], self.parts[r][: self.count[r]])
        return out

    def reduce_frame(self, image, rank, dst=0):
        """reduce(sum) of the zero-masked copy of `image` (own tiles, zeros elsewhere).  `image` is not modified.  Returns the
        assembled frame on `dst` (a buffer owned by this object, overwritten by the next call), None elsewhere."""
        if self.masked is None:
            self.masked = torch.empty_like(image)
            self.zero = torch.zeros((), dtype=image.dtype, device=image.device)
        torch.where(self.mask[rank].unsqueeze(-1), image, self.zero, out=self.masked)      # (a select, not a multiply: foreign pixels may hold anything)
        if self.world > 1:
            dist.reduce(self.masked, dst=dst, op=dist.ReduceOp.SUM)
        return self.masked if rank == dst else None
