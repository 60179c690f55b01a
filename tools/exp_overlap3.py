#!/usr/bin/env python3
"""Experiment (GPU box): the 1 GiB batch cut into n equal parts, each compress -> decompress on its own stream.
usage: exp_overlap3.py n [stagger]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from csnappy_amd import api

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
block, p, nb = 65536, 16, 16384
d_in = api.generate(0, 0xC5A90001, 0, nb, block)


def part(lo, hi):
    m = hi - lo
    b = api.Batch([block] * m)
    return dict(src=d_in[lo * block:hi * block], b=b, out=torch.zeros(b.out_bytes, dtype=torch.uint8, device="cuda"),
                back=torch.zeros(m * block, dtype=torch.uint8, device="cuda"),
                cap=torch.full((m,), block, dtype=torch.int32, device="cuda"),
                st=torch.zeros(m, dtype=torch.int32, device="cuda"), pr=torch.zeros(m, dtype=torch.int32, device="cuda"))


def comp(P):
    b = P["b"]
    api.compress_batch(P["src"], b.d_in_off, b.d_in_len, b.max_in_len, P["out"], b.d_out_off, b.d_out_len, p, 0, b.d_ws)


def dec(P):
    b = P["b"]
    api.decompress_batch(P["out"], b.d_out_off, b.d_out_len, P["back"], b.d_in_off, P["cap"], P["st"], P["pr"], 0)


whole = part(0, nb)
cuts = [nb * i // n for i in range(n + 1)]
parts = [part(cuts[i], cuts[i + 1]) for i in range(n)]
streams = [torch.cuda.Stream() for _ in range(n)]


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
    comp(whole); dec(whole)


def multi():
    cur = torch.cuda.current_stream()
    for s in streams:
        s.wait_stream(cur)
    for P, s in zip(parts, streams):
        with torch.cuda.stream(s):
            comp(P); dec(P)
    for s in streams:
        cur.wait_stream(s)


serial(); multi(); torch.cuda.synchronize()
assert torch.equal(whole["back"], d_in) and all(torch.equal(P["back"], P["src"]) for P in parts)
ts, tm = timed(serial), timed(multi)
print(f"{n} parts on {n} streams: one stream {ts:.2f} ms ({1e3 / ts:.1f} GiB/s), parts {tm:.2f} ms ({1e3 / tm:.1f} GiB/s)")
