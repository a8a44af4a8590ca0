"""The N > 1 exchange step on CPU: two processes over gloo gather their tile-compact slabs to
rank 0, which rebuilds the frame. The same gather_slabs() runs over RCCL in bench.py."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(rank, world, port, w, h, result_path):
    from raytracing_amd import distributed as rd

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tx, ty, shard, slot, counts = rd.tile_layout(w, h, world)
    slab = int(counts.max())
    # Each rank "renders" its own tiles: pixel value encodes (x, y) so that any misplacement shows.
    yy, xx = np.meshgrid(np.arange(ty * 32), np.arange(tx * 32), indexing="ij")
    full = np.stack([xx % 251, yy % 241, (xx + yy) % 239, np.full_like(xx, 255)], axis=-1).astype(np.uint8)
    mine = np.zeros((slab, 32, 32, 4), dtype=np.uint8)
    for t in np.flatnonzero(shard == rank):
        x0, y0 = (t % tx) * 32, (t // tx) * 32
        mine[slot[t]] = full[y0:y0 + 32, x0:x0 + 32]
    compact = torch.from_numpy(mine.reshape(-1).copy())
    gathered, _ = rd.gather_slabs(compact, slab * rd.TILE_BYTES, dst=0)
    if rank == 0:
        frame = rd.deinterleave_numpy(gathered.numpy(), w, h, world)
        np.save(result_path, np.array_equal(frame, full[:h, :w]))
    else:
        assert gathered is None
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gather_rebuilds_the_frame(tmp_path):
    out = str(tmp_path / "ok.npy")
    mp.spawn(_worker, args=(2, _free_port(), 200, 120, out), nprocs=2, join=True)
    assert bool(np.load(out))
