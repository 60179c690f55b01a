#!/usr/bin/env python3
"""Development: parse kernel ms per GiB of urls.10K-replicated (GPU box); CSNAPPY_AMD_LIB selects a variant build."""
import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from csnappy_amd import api
nb = 16384
raw = np.fromfile("tests/golden/urls.10K", dtype=np.uint8)
d_in = torch.from_numpy(np.resize(raw, nb * 65536)).cuda()
b = api.Batch([65536] * nb)
d_out = torch.zeros(b.out_bytes, dtype=torch.uint8, device="cuda")
api.lib().csnappy_hip_set_kernel_timing(1)
for it in range(4):
    api.compress_batch(d_in, b.d_in_off, b.d_in_len, b.max_in_len, d_out, b.d_out_off, b.d_out_len, 16, 0, b.d_ws)
    torch.cuda.synchronize()
ms = (C.c_float * 4)(); ln = (C.c_uint32 * 4)()
api.lib().csnappy_hip_get_kernel_timing(ms, ln)
print(os.environ.get("CSNAPPY_AMD_LIB", "default")[-28:], "urls parse ms", round(ms[0] / max(ln[0],1), 3))
