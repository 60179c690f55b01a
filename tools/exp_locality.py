#!/usr/bin/env python3
"""Experiment (GPU box): parser time when the batch's blocks alias K distinct input blocks --
isolates how much of the parser's time is the candidate gathers missing L2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from csnappy_amd import api
block, nb, p = 65536, 16384, 16
d_in = api.generate(api.WG_TEXT, 0xC5A90001, 0, nb, block)
b = api.Batch([block] * nb)
d_out = torch.zeros(b.out_bytes, dtype=torch.uint8, device="cuda")
for K in (nb, 4096, 1024, 256, 64, 16, 1):
    off = ((torch.arange(nb, dtype=torch.int64, device="cuda") % K) * block).contiguous()
    for it in range(3):
        if it == 1:
            api.get_kernel_timing(); api.set_kernel_timing(True)
        api.compress_batch(d_in, off, b.d_in_len, b.max_in_len, d_out, b.d_out_off, b.d_out_len, p, api.STREAM, b.d_ws)
    torch.cuda.synchronize()
    api.set_kernel_timing(False)
    kt = api.get_kernel_timing()
    print(f"K={K:6d} distinct blocks ({K*block/2**20:8.1f} MiB of input): " + " ".join(f"{k.replace('snappy_','')}={ms/max(c,1):.3f}ms" for k,(ms,c) in sorted(kt.items()) if c), flush=True)
