"""Block-range sharding of a batch across ranks (one process per GPU) and the only collective
the path has: gathering the per-rank compacted streams.

Blocks (and the 32 KiB fragments inside them) share no state (reference
csnappy_compress.c:76-84, per-fragment memset :501, offsets relative to the fragment base
:479,546,550), so rank r of R simply takes the contiguous block range [r*B/R, (r+1)*B/R): the
compress and decompress kernels need no communication.  RCCL (torch.distributed "nccl") is used
only to assemble the final stream: an all_gather of the per-rank byte counts followed by an
all_gather of the compacted per-rank streams, padded to the largest.
"""
import time

import numpy as np


def block_range(total_blocks, rank, world):
    """-> (first_block, count) of rank's contiguous share; ranges tile [0, total_blocks)."""
    lo = total_blocks * rank // world
    hi = total_blocks * (rank + 1) // world
    return lo, hi - lo


def dense_offsets(out_len):
    """Exclusive scan of the per-block compressed lengths (torch tensor, any device) -> int64."""
    import torch
    lens = out_len.to(torch.int64)
    off = torch.cumsum(lens, 0) - lens
    return off, int(lens.sum().item())


def compact(d_out, d_out_off, d_out_len):
    """Pack the slot-strided compress output into one dense stream on the device.
    -> (dense uint8 tensor, int64 dense offsets)"""
    import torch
    from . import api
    off, total = dense_offsets(d_out_len)
    dense = torch.empty(max(total, 1), dtype=torch.uint8, device=d_out.device)
    api.compact_batch(d_out, d_out_off, d_out_len, off, dense)
    return dense[:total], off


def gather_streams(dense, dist, world, group=None):
    """all_gather variable-length per-rank streams.  Works on any backend (nccl on GPUs, gloo in
    the CPU tests).  -> (list of per-rank uint8 tensors, sizes list)"""
    import torch
    size = torch.tensor([dense.numel()], dtype=torch.int64, device=dense.device)
    sizes = [torch.zeros_like(size) for _ in range(world)]
    dist.all_gather(sizes, size, group=group)
    sizes = [int(s.item()) for s in sizes]
    pad = max(max(sizes), 1)
    mine = torch.zeros(pad, dtype=torch.uint8, device=dense.device)
    mine[:dense.numel()] = dense
    bufs = [torch.empty(pad, dtype=torch.uint8, device=dense.device) for _ in range(world)]
    dist.all_gather(bufs, mine, group=group)
    return [bufs[r][:sizes[r]] for r in range(world)], sizes


def time_gather_compacted(d_out, b, dist, world, reps=3):
    """Time compaction + the RCCL gather of the final stream (reported next to, never inside,
    the codec throughput)."""
    import torch
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        dense, _ = compact(d_out, b.d_out_off, b.d_out_len)
        parts, sizes = gather_streams(dense, dist, world)
    torch.cuda.synchronize()
    dist.barrier()
    dt = (time.perf_counter() - t0) / reps
    total = int(np.sum(sizes))
    return {"ms": round(dt * 1e3, 3), "gathered_bytes": total,
            "GBps": round(total / dt / 1e9, 3), "what": "compact + all_gather of per-rank streams"}
