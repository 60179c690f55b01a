#!/usr/bin/env python3
"""Development: decompress kernel ms per GiB of G_text (GPU box); CSNAPPY_AMD_LIB selects a variant build."""
import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
import torch
from csnappy_amd import api
nb = 16384
d_in = api.generate(0, 0xC5A90001, 0, nb, 65536)
b = api.Batch([65536] * nb)
d_out = torch.zeros(b.out_bytes, dtype=torch.uint8, device="cuda")
api.compress_batch(d_in, b.d_in_off, b.d_in_len, b.max_in_len, d_out, b.d_out_off, b.d_out_len, 16, 0, b.d_ws)
d_back = torch.zeros(nb * 65536, dtype=torch.uint8, device="cuda")
back_off = torch.arange(nb, dtype=torch.int64, device="cuda") * 65536
cap = torch.full((nb,), 65536, dtype=torch.int32, device="cuda")
status = torch.zeros(nb, dtype=torch.int32, device="cuda"); produced = torch.zeros(nb, dtype=torch.int32, device="cuda")
ts = []
for it in range(5):
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); api.decompress_batch(d_out, b.d_out_off, b.d_out_len, d_back, back_off, cap, status, produced, 0); t1.record()
    torch.cuda.synchronize(); ts.append(t0.elapsed_time(t1))
print(os.environ.get("CSNAPPY_AMD_LIB", "default")[-28:], "decompress ms per GiB", round(sorted(ts)[1], 3), "equal" if torch.equal(d_back, d_in.view(-1)) else "DIFFERENT (timing build)")
