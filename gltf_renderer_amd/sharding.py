"""Pixel-tile sharding of one frame across the GPUs of a node (new capability).

The reference is single-GPU.  Every pixel-sample is independent (its RNG stream is keyed by
(pixel, seed, counter), PathTracer.lib.hlsl:146), so a frame shards with no data-path exchange:
16x16 tiles in row-major order, tile t belongs to rank t % N (interleaved for load balance), each rank
renders into a zeroed full-size RGBA32F image, and ONE collective per frame assembles it: reduce(sum)
to rank 0 -- tiles are disjoint, so the sum is a gather and the result is bit-identical to a 1-rank frame.
The same function runs over RCCL on GPUs (backend "nccl") and over gloo in the CPU tests.
"""
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
