"""Block-range sharding of a batch across ranks (one process per GPU) and the only collective
the path has: gathering the per-rank compacted streams.

Blocks (and the 32 KiB fragments inside them) share no state (reference
csnappy_compress.c:76-84, per-fragment memset :501, offsets relative to the fragment base
:479,546,550), so rank r of R simply takes the contiguous block range [r*B/R, (r+1)*B/R): the
compress and decompress kernels need no communication.  RCCL (torch.distributed "nccl") is used
only to assemble the final stream: an all_gather of the per-rank byte counts (8 B
per rank), then either
  * gather_to_root   grouped send/recv: every rank sends its compacted stream once, straight into
                     its place in root's output buffer (no padding, no staging copy; over xGMI the
                     seven peers use seven different links to root), or
  * gather_streams   an all_gather padded to the largest stream, when every rank needs the whole
                     stream.
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


def compact(d_out, d_out_off, d_out_len, into=None):
    """Pack the slot-strided compress output into one dense stream on the device.
    into: a uint8 tensor of at least the stream's size to pack into (no allocation then).
    -> (dense uint8 tensor, int64 dense offsets)"""
    import torch
    from . import api
    off, total = api.dense_offsets(d_out_len) if d_out_len.is_cuda else dense_offsets(d_out_len)
    if into is not None and into.numel() >= max(total, 1):
        dense = into
    else:
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


def exchange_sizes(dense, dist, world, group=None):
    """all_gather of the per-rank stream sizes (8 B per rank) -> list of ints."""
    import torch
    size = torch.tensor([dense.numel()], dtype=torch.int64, device=dense.device)
    sizes = [torch.zeros_like(size) for _ in range(world)]
    dist.all_gather(sizes, size, group=group)
    return [int(s.item()) for s in sizes]


def root_buffer(sizes, device):
    """The buffer `root` assembles the streams in (the one allocation of the gather that only one
    rank makes: callers that must not hang make it BEFORE the ranks agree to enter the exchange)."""
    import torch
    return torch.empty(max(int(np.sum(sizes)), 1), dtype=torch.uint8, device=device)


def gather_to_root(dense, dist, world, root=0, group=None, sizes=None, out=None):
    """Variable-size gather of the per-rank streams to `root` with grouped point-to-point
    operations (ncclGroupStart/End under the nccl backend).  The size exchange is the only host
    synchronisation (skipped when the caller passes `sizes`); `out`: root's buffer, if the caller
    made it already (root_buffer).  -> (assembled uint8 tensor on root / None elsewhere, sizes list)"""
    rank = dist.get_rank(group)
    if sizes is None:
        sizes = exchange_sizes(dense, dist, world, group)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    if rank == root:
        if out is None or out.numel() < max(int(offs[-1]), 1):
            out = root_buffer(sizes, dense.device)
        out[int(offs[root]):int(offs[root + 1])] = dense
        ops = [dist.P2POp(dist.irecv, out[int(offs[r]):int(offs[r + 1])], r, group)
               for r in range(world) if r != root and sizes[r] > 0]
    else:
        out = None
        ops = [dist.P2POp(dist.isend, dense, root, group)] if dense.numel() > 0 else []
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return (out[:int(offs[-1])] if out is not None else None), sizes


def gather_lengths(lens, counts, dist, world, group=None):
    """all_gather of the per-block compressed lengths when ranks hold different numbers of blocks
    (block ranges of a batch that does not divide by the world size): padded to the largest count
    -- B x 4 bytes in total, tiny next to the payload.  -> one int64 tensor in block order."""
    import torch
    pad = max(max(counts), 1)
    mine = torch.zeros(pad, dtype=torch.int64, device=lens.device)
    mine[:lens.numel()] = lens.to(torch.int64)
    bufs = [torch.empty(pad, dtype=torch.int64, device=lens.device) for _ in range(world)]
    dist.all_gather(bufs, mine, group=group)
    return torch.cat([bufs[r][:counts[r]] for r in range(world)])


def time_gather_compacted(d_out, b, dist, world, reps=3, cnt=None, dense=None, root_out=None):
    """Time compaction + the RCCL gather of the final stream (reported next to, never inside,
    the codec throughput).  cnt: the blocks of `b` that the last launch filled (a batch's last
    chunk may be shorter than the descriptors; the lengths behind it are stale).
    dense / root_out: buffers made beforehand (this rank's stream, root's assembly buffer): the
    timed loop then allocates nothing -- it packs into `dense` and receives into `root_out`."""
    import torch
    cnt = len(b.d_out_len) if cnt is None else cnt
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        dense, _ = compact(d_out, b.d_out_off[:cnt], b.d_out_len[:cnt], into=dense)
        _, sizes = gather_to_root(dense, dist, world, out=root_out)
    torch.cuda.synchronize()
    dist.barrier()
    dt = (time.perf_counter() - t0) / reps
    total = int(np.sum(sizes))
    return {"ms": round(dt * 1e3, 3), "gathered_bytes": total,
            "GBps": round(total / dt / 1e9, 3),
            "what": "compact + size exchange + grouped send/recv of the per-rank streams to rank 0"}
