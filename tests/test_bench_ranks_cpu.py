"""bench.py's rank / world path on CPU: two gloo ranks run bench.run() with an engine in which the
CPU checker stands in for the HIP kernels (bench.GpuEngine is the product path; this file is a
test of the plumbing around it -- block ranges per rank, the slowest-rank reduction of the wall
and per-kernel times, n_ranks_seen, the default-on gather of the compacted streams to rank 0 and
the record's shape -- so that a first 8-GPU driver run does not die in it)."""
import hashlib
import os
import socket
import sys
import time

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import bench  # noqa: E402
import oracle  # noqa: E402
from csnappy_amd import api, shard  # noqa: E402

BLOCK, NB = 65536, 24  # blocks per rank


class OracleEngine:
    """Same shape as bench.GpuEngine; tensors live on the CPU, the oracle does the codec work."""

    def __init__(self):
        self.torch = torch
        self.device = torch.device("cpu")
        self.codec = oracle.Port()
        self.on = False
        self.ms = {}

    def generate(self, kind, seed, first, nb, block, urls=None):
        return torch.from_numpy(api.generate_host(kind, seed, first, nb, block).copy())

    def batch(self, lens):
        return api.Batch(lens, device="cpu")

    def zeros(self, n, dtype):
        return torch.zeros(n + 64, dtype=dtype)[:n] if dtype == torch.uint8 else torch.zeros(n, dtype=dtype)

    def full(self, n, value, dtype):
        return torch.full((n,), value, dtype=dtype)

    def _clock(self, name, t0):
        if self.on:
            ms, c = self.ms.get(name, (0.0, 0))
            self.ms[name] = (ms + (time.perf_counter() - t0) * 1e3, c + 1)

    def compress(self, src, b, cnt, d_out, p, mode):
        t0 = time.perf_counter()
        _, out_len = oracle.batch_compress(self.codec, src.numpy(), b.in_off[:cnt], b.in_len[:cnt], b.out_off[:cnt],
                                           d_out.numel() - 64, p, mode, out=d_out.numpy())
        b.d_out_len[:cnt] = torch.from_numpy(out_len.astype(np.int32))
        self._clock("snappy_parse_fragments", t0)
        self._clock("snappy_emit_blocks", time.perf_counter())

    def decompress(self, d_out, b, cnt, d_back, cap, status, produced, mode):
        t0 = time.perf_counter()
        lens = b.d_out_len[:cnt].numpy().astype(np.uint32)
        back = np.zeros(d_back.numel() + 64, dtype=np.uint8)
        _, st, pr = oracle.batch_decompress(self.codec, d_out.numpy(), b.out_off[:cnt], lens, b.in_off[:cnt],
                                            cap[:cnt].numpy().astype(np.uint32), d_back.numel(), mode, out=back)
        d_back[:] = torch.from_numpy(back[:d_back.numel()])
        status[:cnt] = torch.from_numpy(st)
        produced[:cnt] = torch.from_numpy(pr.astype(np.int32))
        self._clock("snappy_decompress_blocks", t0)

    def sync(self):
        pass

    def timing(self, on):
        self.on = on
        if on:
            self.ms = {}

    def kernel_times(self):
        return {k: self.ms.get(k, (0.0, 0)) for k in
                ("snappy_parse_fragments", "snappy_emit_blocks", "snappy_decompress_blocks")}

    def copy_bandwidth(self):
        return 1.0

    def verify(self, *a):
        return None

    def time_gather(self, d_out, b, dist_, world):
        # compact on the host (api.compact_batch is a kernel), then the product's own gather
        t0 = time.perf_counter()
        out, lens = d_out.numpy(), b.d_out_len.numpy()
        dense = torch.from_numpy(np.concatenate([out[int(o):int(o) + int(n)] for o, n in zip(b.out_off, lens)]))
        rooted, sizes = shard.gather_to_root(dense, dist_, world)
        self.rooted = rooted
        return {"ms": round((time.perf_counter() - t0) * 1e3, 3), "gathered_bytes": int(np.sum(sizes)), "GBps": 0.0,
                "what": "test stand-in"}


def _args(world):
    return bench.parse_args(["--gpus", str(world), "--steps", "2", "--warmup", "1", "--gib", str(NB * BLOCK / 2 ** 30),
                             "--no-cpu-baseline", "--verify-gib", "0"])


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
        eng = OracleEngine()
        rec = bench.run(_args(world), eng, dist, rank, world)
        if rank == 0:
            ret["rec"] = rec
            ret["gathered_sha"] = hashlib.sha256(eng.rooted.numpy().tobytes()).hexdigest()
        else:
            assert rec is None
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_through_bench_run():
    world = 2
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        rec, gathered = ret["rec"], ret["gathered_sha"]
    assert rec["n_gpus"] == 2 and rec["n_ranks_seen"] == 2 and rec["scaling"] == "weak"
    assert rec["config"]["blocks_per_gpu"] == NB and rec["value"] > 0
    assert rec["rank_ms_per_step"]["min"] <= rec["rank_ms_per_step"]["max"] == rec["ms_per_step"]
    assert set(rec["kernels"]) == {"snappy_parse_fragments", "snappy_emit_blocks", "snappy_decompress_blocks"}
    assert rec["kernels"]["snappy_decompress_blocks"]["launches"] == 2  # one chunk per step, two steps
    assert rec["roofline"]["operation"] in ("compress", "decompress") and rec["roofline"]["frac"] > 0
    # more than one rank: the gather of the final stream is timed by default, next to `value`
    assert rec["gather"]["gathered_bytes"] > 0 and 0 < rec["value_with_gather"] < rec["value"]
    assert "cpu_baseline" not in rec  # rank 0 at N = 1 only
    # the stream rank 0 assembled = the blocks of both ranks' ranges, in block order
    kind, seed, block, p, mode, _ = bench.WORKLOADS["text"]
    host = api.generate_host(kind, seed, 0, world * NB, block)
    want = b"".join(oracle.Port().compress_blocks(host, block, p, mode))
    assert rec["gather"]["gathered_bytes"] == len(want) and gathered == hashlib.sha256(want).hexdigest()


def test_one_rank_without_a_process_group():
    rec = bench.run(_args(1), OracleEngine(), None, 0, 1)
    assert rec["n_gpus"] == 1 and rec["n_ranks_seen"] == 1 and "gather" not in rec and rec["vs_baseline"] is None
    assert rec["metric"] == "GiB/s compress+decompress on 64KiB blocks" and rec["unit"] == "GiB/s"
    assert 0.3 < rec["compressed_ratio"] < 0.7
