"""N>1 path on CPU: world_size-2 gloo run of the sharding + gather logic in csnappy_amd/shard.py.
Each rank produces its block range with the oracle standing in for the GPU kernels (this is a
test of the sharding/gather plumbing, which is backend-agnostic), gathers the compacted streams,
and rank 0 checks the assembled stream equals the single-process result."""
import hashlib
import json
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from csnappy_amd import api, shard

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "golden.json")))


def test_block_ranges_tile_the_batch():
    for total in (0, 1, 7, 16384, 1048576, 4194304 + 3):
        for world in (1, 2, 3, 4, 8):
            parts = [shard.block_range(total, r, world) for r in range(world)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == total
            for (a, ca), (b2, _) in zip(parts, parts[1:]):
                assert a + ca == b2
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1


def test_dense_offsets():
    lens = torch.tensor([5, 0, 7, 1], dtype=torch.int32)
    off, total = shard.dense_offsets(lens)
    assert off.tolist() == [0, 5, 5, 12] and total == 13


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = GOLD["workloads"]["G_text_64k_p16"]
        first, count = shard.block_range(g["nblocks"], rank, world)
        counts = [shard.block_range(g["nblocks"], r, world)[1] for r in range(world)]
        # any block range can be generated independently on any rank
        host = api.generate_host(g["kind"], g["seed"], first, count, g["block"])
        outs = oracle.Port().compress_blocks(host, g["block"], g["p"], g["mode"])
        dense = torch.from_numpy(np.frombuffer(b"".join(outs), dtype=np.uint8).copy())
        parts, sizes = shard.gather_streams(dense, dist, world)            # everyone gets everything
        rooted, sizes2 = shard.gather_to_root(dense, dist, world, root=0)  # grouped send/recv to rank 0
        lens = torch.tensor([len(o) for o in outs], dtype=torch.int64)
        all_lens = shard.gather_lengths(lens, counts, dist, world)         # uneven ranges too
        assert (rooted is None) == (rank != 0)
        if rank == 0:
            stream = b"".join(bytes(p.numpy()) for p in parts)
            ret["sha"] = hashlib.sha256(stream).hexdigest()
            ret["sha_rooted"] = hashlib.sha256(bytes(rooted.numpy())).hexdigest()
            ret["sizes"] = sizes
            ret["sizes_rooted"] = sizes2
            ret["lens"] = all_lens.tolist()
            ret["counts"] = counts
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_reassembles_the_single_process_stream(world):
    """64 blocks over 2 ranks (even) and over 3 ranks (21 / 21 / 22 blocks): both gathers give the
    unsharded stream, and the per-block lengths arrive in block order."""
    g = GOLD["workloads"]["G_text_64k_p16"]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert ret["sha"] == g["sha256"]          # rank-order concatenation == unsharded stream
    assert ret["sha_rooted"] == g["sha256"]
    assert sum(ret["sizes"]) == sum(g["lens"]) and ret["sizes_rooted"] == ret["sizes"]
    assert ret["lens"] == g["lens"]
    assert sum(ret["counts"]) == g["nblocks"] and (world != 3 or len(set(ret["counts"])) == 2)
