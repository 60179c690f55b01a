"""Development: parser time when every block of the batch is the SAME 64 KiB of text (the candidates' lines are shared by all
fragments in flight: what the parser would do if its gathers hit the L2)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.getcwd())
import torch
from csnappy_amd import api

nb = 16384
d_in = api.generate(0, 0xC5A90001, 0, nb, 65536)
same = int(sys.argv[1]) if len(sys.argv) > 1 else 1
if same:
    v = d_in[: nb * 65536].view(nb, 65536)
    v[:] = v[:same].repeat(nb // same, 1)
b = api.Batch([65536] * nb)
d_out = torch.zeros(b.out_bytes, dtype=torch.uint8, device="cuda")
L = api.lib()
for it in range(3):
    api.compress_batch(d_in, b.d_in_off, b.d_in_len, b.max_in_len, d_out, b.d_out_off, b.d_out_len, 16, 0, b.d_ws)
torch.cuda.synchronize()
L.csnappy_hip_set_kernel_timing(1)
for it in range(5):
    api.compress_batch(d_in, b.d_in_off, b.d_in_len, b.max_in_len, d_out, b.d_out_off, b.d_out_len, 16, 0, b.d_ws)
torch.cuda.synchronize()
ms = (C.c_float * 4)()
ln = (C.c_uint32 * 4)()
L.csnappy_hip_get_kernel_timing(ms, ln)
print("distinct blocks %d: parse %.3f emit %.3f ms per GiB" % (same if same else nb, ms[0] / 5, ms[1] / 5))
