#!/usr/bin/env python3
"""Debug (GPU box): incompressible 64 KiB blocks -- the long-literal paths of both kernels."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from csnappy_amd import api
# incompressible 64 KiB blocks: decompress = long literals
nb, block = 8192, 65536
d_in = torch.randint(0, 256, (nb * block,), dtype=torch.uint8, device="cuda")
b = api.Batch([block] * nb)
d_out = torch.zeros(b.out_bytes, dtype=torch.uint8, device="cuda")
api.compress_batch(d_in, b.d_in_off, b.d_in_len, b.max_in_len, d_out, b.d_out_off, b.d_out_len, 16, 0, b.d_ws)
back = torch.zeros(nb * block + 64, dtype=torch.uint8, device="cuda")
status = torch.zeros(nb, dtype=torch.int32, device="cuda"); prod = torch.zeros(nb, dtype=torch.int32, device="cuda")
cap = torch.full((nb,), block, dtype=torch.int32, device="cuda")
best = 1e9
for it in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    api.decompress_batch(d_out, b.d_out_off, b.d_out_len, back, b.d_in_off, cap, status, prod, 0)
    torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
assert (status == 0).all().item() and torch.equal(back[:nb*block], d_in)
print(f"random 64 KiB blocks: decompress {nb*block/2**30/best:.1f} GiB/s")
best = 1e9
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    api.compress_batch(d_in, b.d_in_off, b.d_in_len, b.max_in_len, d_out, b.d_out_off, b.d_out_len, 16, 0, b.d_ws)
    torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
print(f"random 64 KiB blocks: compress {nb*block/2**30/best:.1f} GiB/s")
