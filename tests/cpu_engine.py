"""An engine of bench.GpuEngine's shape in which the CPU checker stands in for the HIP kernels:
TEST INFRASTRUCTURE (tests/test_bench_ranks_cpu.py drives bench.run() and `bench.py --gpus 2
--engine tests.cpu_engine:OracleEngine` with it on gloo ranks; bench.py itself never imports
the oracle for anything it measures)."""
import os
import time

import numpy as np
import torch

import oracle
from csnappy_amd import api, shard


class OracleEngine:
    """Same shape as bench.GpuEngine; tensors live on the CPU, the oracle does the codec work."""

    def __init__(self):
        self.torch = torch
        self.device = torch.device("cpu")
        self.codec = oracle.Port()
        self.on = False
        self.ms = {}

    def generate(self, kind, seed, first, nb, block, urls=None):
        if kind < 0:
            rep = np.frombuffer(urls, dtype=np.uint8)
            return torch.from_numpy(rep[(np.arange(nb * block, dtype=np.int64) + first * block) % len(rep)].copy())
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

    def init_dist(self, dist_):
        dist_.init_process_group("gloo")

    def prepare_gather(self, d_out, b, cnt):
        # compact on the host (api.compact_batch is a kernel)
        if os.environ.get("CSNAPPY_TEST_FAIL_PREPARE_ON_RANK") == os.environ.get("RANK", "0"):
            raise MemoryError("injected by the test")
        out, lens = d_out.numpy(), b.d_out_len.numpy()[:cnt]
        return torch.from_numpy(np.concatenate([out[int(o):int(o) + int(n)] for o, n in zip(b.out_off[:cnt], lens)]))

    def root_buffer(self, dense, dist_, world):
        if os.environ.get("CSNAPPY_TEST_FAIL_ROOT_BUFFER") and dist_.get_rank() == 0:
            shard.exchange_sizes(dense, dist_, world)
            raise MemoryError("injected by the test (root buffer)")
        sizes = shard.exchange_sizes(dense, dist_, world)
        return shard.root_buffer(sizes, dense.device) if dist_.get_rank() == 0 else None

    def time_gather(self, dense, root_out, d_out, b, cnt, dist_, world):
        # ... then the product's own gather
        t0 = time.perf_counter()
        rooted, sizes = shard.gather_to_root(dense, dist_, world, out=root_out)
        self.rooted = rooted
        return {"ms": round((time.perf_counter() - t0) * 1e3, 3), "gathered_bytes": int(np.sum(sizes)), "GBps": 0.0,
                "what": "test stand-in"}
