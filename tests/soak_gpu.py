#!/usr/bin/env python3
"""Randomised soak of the HIP path against the oracle (run on the GPU box):
    python tests/soak_gpu.py [seconds]          (clock-seeded, as long as you like)
tests/test_gpu_parity.py::test_soak_slice_compress runs soak(15, seed=20261002) under pytest.
Random batches: mixed block sizes / content classes / table powers / modes / placements; every
block's compressed bytes must equal the checker's and must round-trip."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import oracle  # noqa: E402
from csnappy_amd import api  # noqa: E402

def block(rng, n):
    kind = int(rng.integers(0, 7))
    if kind == 0:
        return rng.integers(0, 256, n, dtype=np.uint8)
    if kind == 1:
        return np.resize(rng.integers(0, int(rng.choice([2, 4, 256])), int(rng.integers(1, 90)), dtype=np.uint8), n)
    if kind == 2:
        return api.generate_host(api.WG_TEXT, int(rng.integers(1, 1 << 30)), int(rng.integers(0, 1000)), 1, max(n, 1))[:n]
    if kind == 3:
        return api.generate_host(api.WG_LOW, int(rng.integers(1, 1 << 30)), int(rng.integers(0, 1000)), 1, max(n, 1))[:n]
    if kind == 4:
        x = np.zeros(n, np.uint8)
        x[::8] = rng.integers(0, int(rng.choice([2, 16, 256])), len(x[::8]), dtype=np.uint8)
        return x
    if kind == 5:
        return np.zeros(n, np.uint8)
    x = rng.integers(0, int(rng.choice([2, 16, 256])), n, dtype=np.uint8)
    for _ in range(8):
        if n > 100:
            s = int(rng.integers(0, n - 50))
            ln = int(rng.integers(4, min(6000, n - s)))
            d = int(rng.integers(0, n - ln))
            x[d:d + ln] = x[s:s + ln].copy()
    return x


def soak(budget, seed=None):
    """Random ragged batches for `budget` seconds; returns (batches, blocks)."""
    chk = oracle.best()
    rng = np.random.default_rng(int(time.time()) if seed is None else seed)
    api.require_device()
    saved = os.environ.get("CSNAPPY_HIP_TABLE")
    try:
        t0, rounds, blocks_done = time.time(), 0, 0
        while time.time() - t0 < budget:
            mode = int(rng.integers(0, 2))
            p = int(rng.integers(9, 17))
            os.environ["CSNAPPY_HIP_TABLE"] = str(rng.choice(["auto", "hash", "dense", "global"]))
            api.reload_knobs()
            nb = int(rng.integers(1, 300))
            top = 32768 if mode else int(rng.choice([300, 5000, 32768, 65536, 200000]))
            lens = [int(rng.choice([0, 1, 14, 15, 16, rng.integers(0, top + 1), top])) for _ in range(nb)]
            xs = [block(rng, n) for n in lens]
            host = np.concatenate(xs) if sum(lens) else np.zeros(0, np.uint8)
            b = api.Batch(lens)
            d_in = torch.from_numpy(np.concatenate([host, np.zeros(16, np.uint8)])).cuda()
            d_out = torch.full((b.out_bytes + 64,), 0xA5, dtype=torch.uint8, device="cuda")
            api.compress_batch(d_in, b.d_in_off, b.d_in_len, b.max_in_len, d_out, b.d_out_off, b.d_out_len, p, mode, b.d_ws)
            d_back = torch.zeros(max(sum(lens), 1), dtype=torch.uint8, device="cuda")
            cap = torch.from_numpy(np.asarray(lens, dtype=np.int32)).cuda()
            status = torch.full((nb,), -99, dtype=torch.int32, device="cuda")
            produced = torch.zeros(nb, dtype=torch.int32, device="cuda")
            api.decompress_batch(d_out, b.d_out_off, b.d_out_len, d_back, b.d_in_off, cap, status, produced, mode)
            torch.cuda.synchronize()
            out, out_len = d_out.cpu().numpy(), b.d_out_len.cpu().numpy().astype(np.uint32)
            assert (out[b.out_bytes:] == 0xA5).all()
            for i, x in enumerate(xs):
                want = chk.compress(x, p) if mode == 0 else chk.compress_fragment(x, p)
                o = int(b.out_off[i])
                got = bytes(out[o:o + int(out_len[i])])
                assert got == want, (rounds, i, len(x), p, mode, os.environ["CSNAPPY_HIP_TABLE"], len(got), len(want))
            assert (status == 0).all().item(), status.cpu().numpy()
            assert np.array_equal(d_back.cpu().numpy()[:sum(lens)], host)
            rounds += 1
            blocks_done += nb
    finally:
        if saved is None:
            os.environ.pop("CSNAPPY_HIP_TABLE", None)
        else:
            os.environ["CSNAPPY_HIP_TABLE"] = saved
        api.reload_knobs()
    return rounds, blocks_done, chk.kind


if __name__ == "__main__":
    t0 = time.time()
    rounds, blocks_done, kind = soak(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0)
    print(f"soak ok: {rounds} batches, {blocks_done} blocks in {time.time() - t0:.0f} s (checker: {kind})")
