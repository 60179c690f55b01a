#!/usr/bin/env python3
"""Randomised soak of the STREAM CALL (csnappy_hip_decompress_stream) on long streams, sound and
damaged (run on the GPU box):
    python tests/soak_stream_gpu.py [seconds]   (clock-seeded)
tests/test_gpu_parity.py::test_soak_slice_stream runs soak(15, seed=20261003) under pytest.
Streams of 100 KiB .. 3 MiB of mixed content from the checker's compressor are decoded as they are,
and after byte flips, cuts, splices of long-literal / far-copy tags, removed or doubled stretches,
and a wrong length header.  Status, produced length and bytes must equal the checker's
csnappy_decompress_noheader on the same body with *dst_len = the header's length; the index must
take sound csnappy streams (fast path) and must never be the reason for a different answer.
Streams whose last tag header is cut off by the end of input are skipped (undefined in the
reference)."""
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
from test_oracle import _body_has_truncated_tag  # noqa: E402

SPLICES = [bytes.fromhex(h) for h in ("f0ff", "f4ffff", "f8ffffff", "fcffffffff", "fc00000080", "ff00000000",
                                      "ff01000000", "fe0000", "fe0100", "0500", "01ff", "fdffff", "fe0090", "ff00900000")]


def content(rng, n):
    parts, have = [], 0
    while have < n:
        kind = int(rng.integers(0, 5))
        m = int(rng.integers(1, 150000))
        if kind == 0:
            part = api.generate_host(api.WG_TEXT, int(rng.integers(1, 1 << 30)), 0, 1, m)
        elif kind == 1:
            part = rng.integers(0, int(rng.choice([2, 16, 256])), m, dtype=np.uint8)
        elif kind == 2:
            part = api.generate_host(api.WG_LOW, int(rng.integers(1, 1 << 30)), 0, 1, m)
        elif kind == 3:
            part = np.resize(rng.integers(0, 256, int(rng.integers(1, 5000)), dtype=np.uint8), m)
        else:  # sparse matches: literals of a few hundred bytes between short copies
            part = rng.integers(0, 256, m, dtype=np.uint8)
            for at in range(0, m - 600, int(rng.integers(200, 900))):
                part[at + 300:at + 300 + 12] = part[at:at + 12]
        parts.append(part)
        have += m
    return np.concatenate(parts)[:n]


def varint(v):
    out = b""
    while v >= 128:
        out += bytes([v & 127 | 128])
        v >>= 7
    return out + bytes([v])


def call(body, ulen):
    d_body = torch.from_numpy(np.frombuffer(body, dtype=np.uint8).copy()).cuda() if body else \
        torch.zeros(0, dtype=torch.uint8, device="cuda")
    d_out = torch.full((ulen + 64,), 0xA5, dtype=torch.uint8, device="cuda")
    st, produced, fast = api.decompress_stream(d_body, ulen, d_out)
    out = d_out.cpu().numpy()
    assert (out[ulen:] == 0xA5).all(), "wrote past the expected length"
    return st, produced, bytes(out[:produced]) if st == 0 else b"", fast


def soak(budget, seed=None):
    """-> (streams checked, of which decoded by fragments, checker kind)"""
    chk = oracle.best()
    rng = np.random.default_rng(int(time.time()) if seed is None else seed)
    api.require_device()
    t0, checked, fasts = time.time(), 0, 0
    while time.time() - t0 < budget:
        x = content(rng, int(rng.integers(100000, 3000000))).tobytes()
        good = chk.compress(x, int(rng.integers(11, 17)))
        hdr, ulen = chk.get_uncompressed_length(good)
        # the sound stream: must go the fast way
        st, produced, out, fast = call(good[hdr:], ulen)
        if not (st == 0 and out == x and fast):
            os.makedirs("gpurun_out", exist_ok=True)
            open("gpurun_out/soak_stream_fail.bin", "wb").write(good)
        assert st == 0 and out == x and fast, ("sound stream", len(x), st, fast)
        checked += 1
        fasts += 1
        for _ in range(int(rng.integers(4, 12))):
            m = bytearray(good[hdr:])
            want_len = ulen
            for _ in range(int(rng.integers(1, 4))):
                k = int(rng.integers(0, 7))
                at = int(rng.integers(0, len(m)))
                if k == 0:
                    m[at] = int(rng.integers(0, 256))
                elif k == 1:
                    m[at:at] = SPLICES[int(rng.integers(0, len(SPLICES)))]
                elif k == 2:
                    del m[int(rng.integers(len(m) // 2, len(m))):]
                elif k == 3:
                    del m[at:at + int(rng.integers(1, 40000))]
                elif k == 4:
                    m[at:at] = m[at:at + int(rng.integers(1, 40000))]
                elif k == 5:
                    want_len = int(rng.choice([ulen - 1, ulen + 1, ulen // 2, ulen + 40000, 1]))
                else:
                    m += bytes(int(rng.integers(1, 9000)))
            body = bytes(m)
            if not body or want_len <= 0 or _body_has_truncated_tag(body):
                continue
            rc, prod, ref = chk.decompress_noheader(body, want_len)
            st, produced, out, fast = call(body, want_len)
            if st != rc or (rc == 0 and (produced != prod or out != ref)):
                os.makedirs("gpurun_out", exist_ok=True)
                open("gpurun_out/soak_stream_fail.bin", "wb").write(varint(want_len) + body)
            assert st == rc, ("status", len(body), want_len, st, rc, fast)
            if rc == 0:
                assert produced == prod and out == ref, ("bytes", len(body), want_len, fast)
            checked += 1
            fasts += fast
    return checked, fasts, type(chk).__name__


if __name__ == "__main__":
    n, f, kind = soak(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0)
    print(f"soak_stream: {n} long streams checked against {kind}, {f} decoded fragment by fragment: all equal")
