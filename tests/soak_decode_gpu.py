#!/usr/bin/env python3
"""Randomised soak of the HIP DECOMPRESSOR on damaged streams (run on the GPU box):
    python tests/soak_decode_gpu.py [seconds]   (clock-seeded)
tests/test_gpu_parity.py::test_soak_slice_decode runs soak(15, seed=20261002) under pytest.
Valid streams from the checker are mutated (byte flips, insertions of long-literal / 4-byte-offset
tags, truncation) and decoded with random dst_len; status, produced length and bytes must equal the
checker's, in STREAM and in FRAGMENT form.  Streams whose tag header is cut off by the end of input
are skipped (undefined in the reference)."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import torch  # noqa: E402
import oracle  # noqa: E402
from csnappy_amd import api  # noqa: E402
from test_oracle import _body_has_truncated_tag, _has_truncated_tag  # noqa: E402
from test_gpu_parity import gpu_decompress  # noqa: E402

def content(rng, n):
    kind = int(rng.integers(0, 4))
    if kind == 0:
        return rng.integers(0, int(rng.choice([2, 16, 256])), n, dtype=np.uint8)
    if kind == 1:
        return np.resize(rng.integers(0, 256, int(rng.integers(1, 70)), dtype=np.uint8), n)
    if kind == 2:
        return api.generate_host(api.WG_TEXT, int(rng.integers(1, 1 << 30)), 0, 1, max(n, 1))[:n]
    return api.generate_host(api.WG_LOW, int(rng.integers(1, 1 << 30)), 0, 1, max(n, 1))[:n]


SPLICES = [bytes.fromhex(h) for h in ("f0ff", "f4ffff", "f8ffffff", "fcffffffff", "fc00000080", "ff00000000",
                                      "ff01000000", "fe0000", "fe0100", "0500", "01ff", "fdffff", "f1", "f2")]

def soak(budget, seed=None):
    """Damaged streams for `budget` seconds; returns (batches, streams, checker kind)."""
    chk, P = oracle.best(), oracle.Port()
    rng = np.random.default_rng(int(time.time()) if seed is None else seed)
    api.require_device()
    t0, rounds, checked = time.time(), 0, 0
    while time.time() - t0 < budget:
        streams, caps = [], []
        for _ in range(int(rng.integers(20, 200))):
            x = content(rng, int(rng.choice([0, 1, 20, rng.integers(0, 3000), rng.integers(0, 70000)])))
            m = bytearray(P.compress(x, int(rng.integers(9, 17))))
            for _ in range(int(rng.integers(0, 4))):
                k = int(rng.integers(0, 4))
                at = int(rng.integers(0, len(m) + 1))
                if k == 0 and len(m):
                    m[min(at, len(m) - 1)] = int(rng.integers(0, 256))
                elif k == 1:
                    m[at:at] = SPLICES[int(rng.integers(0, len(SPLICES)))]
                elif k == 2 and len(m) > 2:
                    del m[-int(rng.integers(1, 3)):]
                elif k == 3 and len(m) > 8:
                    del m[at:at + int(rng.integers(1, 5))]
            s = bytes(m)
            if _has_truncated_tag(s):
                continue
            streams.append(s)
            caps.append(int(rng.choice([len(x), len(x) + 9, max(len(x) - 1, 0), 3 * len(x) + 100, 0])))
        if not streams:
            continue
        st, pr, outs = gpu_decompress(torch, streams, caps, api.STREAM)
        for i, (s, cap) in enumerate(zip(streams, caps)):
            want_rc, _ = chk.decompress(s, cap)
            if st[i] != want_rc:
                os.makedirs("gpurun_out", exist_ok=True)
                open("gpurun_out/soak_decode_fail.txt", "w").write(f"{s.hex()}\n{cap}\n{int(st[i])}\n{want_rc}\n")
            assert st[i] == want_rc, ("stream", s.hex()[:80], cap, int(st[i]), want_rc)
            if want_rc == 0:
                n = P.get_uncompressed_length(s)
                _, prod, body = chk.decompress_noheader(s[n[0]:], cap)
                assert outs[i] == body and pr[i] == prod, ("stream bytes", s.hex()[:80], cap)
        bodies = [s[P.get_uncompressed_length(s)[0]:] if P.get_uncompressed_length(s)[0] > 0 else s for s in streams]
        keep = [i for i, b_ in enumerate(bodies) if len(b_) > 0 and not _body_has_truncated_tag(b_)]
        if keep:
            st2, pr2, outs2 = gpu_decompress(torch, [bodies[i] for i in keep], [caps[i] for i in keep], api.FRAGMENT)
            for j, i in enumerate(keep):
                rc, prod, body = chk.decompress_noheader(bodies[i], caps[i])
                assert st2[j] == rc, ("body", bodies[i].hex()[:80], caps[i], int(st2[j]), rc)
                if rc == 0:
                    assert pr2[j] == prod and outs2[j] == body, ("body bytes", bodies[i].hex()[:80], caps[i])
        rounds += 1
        checked += len(streams)
    return rounds, checked, chk.kind


if __name__ == "__main__":
    t0 = time.time()
    rounds, checked, kind = soak(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0)
    print(f"decode soak ok: {rounds} batches, {checked} damaged streams in {time.time() - t0:.0f} s (checker: {kind})")
