#!/usr/bin/env python3
"""Experiment (GPU box): fill the tails of one launch per kernel with a second, low-priority stream.
The 1 GiB batch is cut in two parts A (share f) and B; A runs on a high-priority stream, B on a low one,
each compress then decompress; compared with the whole batch on one stream.
usage: exp_overlap2.py [f=0.875]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from csnappy_amd import api

f = float(sys.argv[1]) if len(sys.argv) > 1 else 0.875
block, p, nb = 65536, 16, 16384
d_in = api.generate(0, 0xC5A90001, 0, nb, block)


def part(lo, hi):
    n = hi - lo
    b = api.Batch([block] * n)
    return dict(src=d_in[lo * block:hi * block], b=b, out=torch.zeros(b.out_bytes, dtype=torch.uint8, device="cuda"),
                back=torch.zeros(n * block, dtype=torch.uint8, device="cuda"),
                cap=torch.full((n,), block, dtype=torch.int32, device="cuda"),
                st=torch.zeros(n, dtype=torch.int32, device="cuda"), pr=torch.zeros(n, dtype=torch.int32, device="cuda"))


def roundtrip(P):
    b = P["b"]
    api.compress_batch(P["src"], b.d_in_off, b.d_in_len, b.max_in_len, P["out"], b.d_out_off, b.d_out_len, p, 0, b.d_ws)
    api.decompress_batch(P["out"], b.d_out_off, b.d_out_len, P["back"], b.d_in_off, P["cap"], P["st"], P["pr"], 0)


whole = part(0, nb)
cut = int(nb * f)
A, B = part(0, cut), part(cut, nb)
lo_pri, hi_pri = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
s_hi, s_lo = torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)


def timed(fn, reps=8):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


def serial():
    roundtrip(whole)


def two_streams():
    cur = torch.cuda.current_stream()
    s_hi.wait_stream(cur); s_lo.wait_stream(cur)
    with torch.cuda.stream(s_hi):
        roundtrip(A)
    with torch.cuda.stream(s_lo):
        roundtrip(B)
    cur.wait_stream(s_hi); cur.wait_stream(s_lo)


serial(); two_streams(); torch.cuda.synchronize()
assert torch.equal(whole["back"], d_in) and torch.equal(A["back"], d_in[:cut * block]) and torch.equal(B["back"], d_in[cut * block:])
ts, to = timed(serial), timed(two_streams)
print(f"share of the high-priority part {f}: one stream {ts:.2f} ms ({1e3 / ts:.1f} GiB/s), two streams {to:.2f} ms ({1e3 / to:.1f} GiB/s)")
