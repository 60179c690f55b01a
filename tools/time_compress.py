#!/usr/bin/env python3
"""Debug (GPU box): time the compress batch alone, no verification (for kernel experiments whose
output is deliberately wrong).  usage: time_compress.py [text|low|page] [gib]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from csnappy_amd import api

kind, seed, block, p, mode = {"text": (0, 0xC5A90001, 65536, 16, 0), "low": (1, 0xC5A90005, 65536, 16, 0),
                              "page": (2, 0xC5A90004, 4096, 13, 1)}[sys.argv[1] if len(sys.argv) > 1 else "text"]
gib = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
nb = int(gib * (1 << 30)) // block
d_in = api.generate(kind, seed, 0, nb, block)
b = api.Batch([block] * nb)
d_out = torch.zeros(b.out_bytes, dtype=torch.uint8, device="cuda")
best = 1e9
for it in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    api.compress_batch(d_in, b.d_in_off, b.d_in_len, b.max_in_len, d_out, b.d_out_off, b.d_out_len, p, mode, b.d_ws)
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
print(f"compress {gib} GiB {sys.argv[1] if len(sys.argv) > 1 else 'text'}: {best*1e3:.2f} ms  {gib/best:.2f} GiB/s")
