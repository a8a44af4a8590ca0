"""Frame sharding across the GPUs of one node: one process per GPU, every process holds the whole
scene, screen tiles are dealt out diagonally ((tx + ty) % n_shards), and rank 0 collects the
tiles with one gather over RCCL (torch.distributed backend "nccl"; "gloo" on CPU in tests).

The tile-to-shard rule and the slot order (ascending tile id within a shard) are the ones the
frame kernel and rt_deinterleave_tiles_device use (include/rays_hip.h).
"""
import numpy as np

RT_TILE = 32
TILE_BYTES = RT_TILE * RT_TILE * 4


def tile_layout(width, height, n_shards):
    """Returns (tiles_x, tiles_y, shard_of_tile, slot_of_tile, tiles_per_shard)."""
    tx = (width + RT_TILE - 1) // RT_TILE
    ty = (height + RT_TILE - 1) // RT_TILE
    t = np.arange(tx * ty)
    shard = ((t % tx) + (t // tx)) % n_shards
    slot = np.zeros(tx * ty, dtype=np.int64)
    counts = np.zeros(n_shards, dtype=np.int64)
    for s in range(n_shards):
        m = shard == s
        slot[m] = np.arange(m.sum())
        counts[s] = m.sum()
    return tx, ty, shard, slot, counts


def slab_tiles(width, height, n_shards):
    """Tiles per gathered slab: the largest shard's tile count (slabs are equal-sized)."""
    return int(tile_layout(width, height, n_shards)[4].max())


def deinterleave_numpy(gathered, width, height, n_shards):
    """CPU statement of rt_deinterleave_tiles_device: gathered is (n_shards, slab_tiles, 32, 32, 4)
    u8 (or anything reshapeable to it). Returns the (height, width, 4) frame."""
    tx, ty, shard, slot, counts = tile_layout(width, height, n_shards)
    g = np.asarray(gathered, dtype=np.uint8).reshape(n_shards, -1, RT_TILE, RT_TILE, 4)
    frame = np.zeros((ty * RT_TILE, tx * RT_TILE, 4), dtype=np.uint8)
    for t in range(tx * ty):
        x0, y0 = (t % tx) * RT_TILE, (t // tx) * RT_TILE
        frame[y0:y0 + RT_TILE, x0:x0 + RT_TILE] = g[shard[t], slot[t]]
    return frame[:height, :width]


def gather_slabs(compact, slab_bytes, dst=0, async_op=False):
    """One exchange step per frame: every rank contributes its tile-compact buffer (padded to
    slab_bytes), rank `dst` receives (world, slab_bytes). Works for CUDA tensors over RCCL and
    CPU tensors over gloo. Returns (gathered or None, work or None)."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(), dist.get_rank()
    assert compact.numel() == slab_bytes and compact.dtype == torch.uint8
    out = None
    gather_list = None
    if rank == dst:
        out = torch.empty((world, slab_bytes), dtype=torch.uint8, device=compact.device)
        gather_list = list(out.unbind(0))
    work = dist.gather(compact, gather_list, dst=dst, async_op=async_op)
    return out, work
