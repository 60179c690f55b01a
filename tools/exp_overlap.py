#!/usr/bin/env python3
"""Experiment (GPU box): do the parser and the decompressor share a GPU better than they take turns?
Two halves of a 1 GiB batch: serial = compress(A) decompress(A) compress(B) decompress(B) on one stream;
overlapped = compress(B) on one stream while decompress(A) runs on another.
usage: [CSNAPPY_HIP_WGS_PER_CU=14] exp_overlap.py [MiB per half]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from csnappy_amd import api

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 512
block, p = 65536, 16
nb = (mib << 20) // block
halves = []
for h in range(2):
    d_in = api.generate(0, 0xC5A90001, h * nb, nb, block)
    b = api.Batch([block] * nb)
    d_out = torch.zeros(b.out_bytes, dtype=torch.uint8, device="cuda")
    d_back = torch.zeros(nb * block, dtype=torch.uint8, device="cuda")
    cap = torch.full((nb,), block, dtype=torch.int32, device="cuda")
    st = torch.zeros(nb, dtype=torch.int32, device="cuda")
    pr = torch.zeros(nb, dtype=torch.int32, device="cuda")
    halves.append((d_in, b, d_out, d_back, cap, st, pr))


def comp(h):
    d_in, b, d_out, *_ = halves[h]
    api.compress_batch(d_in, b.d_in_off, b.d_in_len, b.max_in_len, d_out, b.d_out_off, b.d_out_len, p, 0, b.d_ws)


def dec(h):
    d_in, b, d_out, d_back, cap, st, pr = halves[h]
    api.decompress_batch(d_out, b.d_out_off, b.d_out_len, d_back, b.d_in_off, cap, st, pr, 0)


for h in range(2):
    comp(h); dec(h)
torch.cuda.synchronize()
assert all(torch.equal(halves[h][3], halves[h][0]) for h in range(2))


def timed(fn, reps=6):
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
    for h in range(2):
        comp(h); dec(h)


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def overlapped():
    # steady state of a two-stage pipeline over halves: decompress(A) beside compress(B), then the roles swapped
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    for a, b_ in ((0, 1), (1, 0)):
        with torch.cuda.stream(s1):
            comp(b_)
        with torch.cuda.stream(s2):
            dec(a)
        s1.wait_stream(s2); s2.wait_stream(s1)
    cur.wait_stream(s1); cur.wait_stream(s2)


ts, to = timed(serial), timed(overlapped)
gib = 2 * mib / 1024
print(f"WGS_PER_CU={os.environ.get('CSNAPPY_HIP_WGS_PER_CU', 'default')}: serial {ts:.2f} ms ({gib / ts * 1e3:.1f} GiB/s round trip), "
      f"overlapped {to:.2f} ms ({gib / to * 1e3:.1f} GiB/s)")
