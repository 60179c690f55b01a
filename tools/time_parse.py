#!/usr/bin/env python3
"""Development: parse kernel ms per GiB, several repetitions (GPU box).
usage: [CSNAPPY_AMD_LIB=...] time_parse.py [text|urls|low|page] [MiB]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from csnappy_amd import api

w = sys.argv[1] if len(sys.argv) > 1 else "text"
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
kind, seed, block, p, mode = {"text": (0, 0xC5A90001, 65536, 16, 0), "low": (1, 0xC5A90005, 65536, 16, 0),
                              "page": (2, 0xC5A90004, 4096, 13, 1), "urls": (-1, 0, 65536, 16, 0)}[w]
nb = (mib << 20) // block
if kind >= 0:
    d_in = api.generate(kind, seed, 0, nb, block)
else:
    raw = np.fromfile(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "urls.10K"),
                      dtype=np.uint8)
    d_in = torch.from_numpy(np.resize(raw, nb * block)).cuda()
b = api.Batch([block] * nb)
d_out = torch.zeros(b.out_bytes, dtype=torch.uint8, device="cuda")
ts = []
for it in range(6):
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    api.compress_batch(d_in, b.d_in_off, b.d_in_len, b.max_in_len, d_out, b.d_out_off, b.d_out_len, p, mode, b.d_ws)
    t1.record()
    torch.cuda.synchronize()
    ts.append(t0.elapsed_time(t1))
ts = sorted(ts[1:])
print(f"{os.environ.get('CSNAPPY_AMD_LIB', 'default').split('/')[-2] if os.environ.get('CSNAPPY_AMD_LIB') else 'default':10s} {w} {mib} MiB: compress (parse+emit) median {ts[len(ts)//2] * 1024 / mib:.3f} min {ts[0] * 1024 / mib:.3f} ms per GiB")
