"""Image-tile sharding across ranks (one process per GPU) and the gather of the per-rank accumulators.

The image is cut into 8x8-pixel blocks; block b (row-major over blocks) belongs to rank b % world; a rank's compact
accumulator is [local_block][py][px][rgb] fp32 with local_block = b // world, padded to the size of rank 0's so that the
gather has equal counts (include/solstrale_hip.h, sol_scene_set_partition). On the GPU the ownership test lives in the
render kernel and the inverse permutation in sol_unpermute; the numpy functions below restate both for the CPU (gloo)
tests of the N>1 path and document the layout.
"""
import numpy as np

TILE = 8


def block_grid(width, height):
    return (width + TILE - 1) // TILE, (height + TILE - 1) // TILE


def accum_floats(width, height, world):
    bx, by = block_grid(width, height)
    return ((bx * by + world - 1) // world) * 64 * 3


def _slots(width, height, world):
    """Per pixel (row-major, row 0 = top): owning rank and the slot inside that rank's compact buffer."""
    bx, _ = block_grid(width, height)
    y, x = np.mgrid[0:height, 0:width]
    b = (y // TILE) * bx + (x // TILE)
    return b % world, (b // world) * 64 + (y % TILE) * TILE + (x % TILE)


def compact_from_image(image, rank, world):
    """The compact buffer rank `rank` holds when the full image is `image` (H, W, 3)."""
    h, w, _ = image.shape
    owner, slot = _slots(w, h, world)
    out = np.zeros(accum_floats(w, h, world), dtype=image.dtype).reshape(-1, 3)
    m = owner == rank
    out[slot[m]] = image[m]
    return out.reshape(-1)


def image_from_gathered(gathered, width, height, world):
    """Inverse permutation: `gathered` = world compact buffers back to back (what sol_unpermute does on the device)."""
    owner, slot = _slots(width, height, world)
    n = accum_floats(width, height, world) // 3
    g = np.asarray(gathered).reshape(world, n, 3)
    return g[owner, slot]


def gather_to_rank0(local, world, rank, group=None):
    """RCCL (or gloo) gather of equal-sized compact accumulators (torch tensors) to rank 0; returns the concatenated
    tensor on rank 0, None elsewhere. One collective per emitted image, never per pass."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return local
    if rank == 0:
        out = torch.empty(world * local.numel(), dtype=local.dtype, device=local.device)
        parts = list(out.split(local.numel()))
        dist.gather(local, gather_list=parts, dst=0, group=group)
        return out
    dist.gather(local, gather_list=None, dst=0, group=group)
    return None
