"""Development: ms per GiB of the decompress kernel on a config (default G_text), by the library's own kernel timers."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.getcwd())
import torch
from csnappy_amd import api

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
nb = 16384
d_in = api.generate(cfg, 0xC5A90001, 0, nb, 65536)
b = api.Batch([65536] * nb)
d_out = torch.zeros(b.out_bytes, dtype=torch.uint8, device="cuda")
L = api.lib()
api.compress_batch(d_in, b.d_in_off, b.d_in_len, b.max_in_len, d_out, b.d_out_off, b.d_out_len, 16, 0, b.d_ws)
torch.cuda.synchronize()
d_back = torch.zeros(nb * 65536, dtype=torch.uint8, device="cuda")
d_cap = torch.full((nb,), 65536, dtype=torch.int32, device="cuda")
d_boff = torch.arange(nb, dtype=torch.int64, device="cuda") * 65536
d_status = torch.zeros(nb, dtype=torch.int32, device="cuda")
d_prod = torch.zeros(nb, dtype=torch.int32, device="cuda")
for it in range(2):
    api.decompress_batch(d_out, b.d_out_off, b.d_out_len, d_back, d_boff, d_cap, d_status, d_prod, 0)
torch.cuda.synchronize()
assert torch.equal(d_back, d_in[: nb * 65536]) and int(d_status.abs().sum()) == 0
L.csnappy_hip_set_kernel_timing(1)
for it in range(5):
    api.decompress_batch(d_out, b.d_out_off, b.d_out_len, d_back, d_boff, d_cap, d_status, d_prod, 0)
torch.cuda.synchronize()
ms = (C.c_float * 4)()
ln = (C.c_uint32 * 4)()
L.csnappy_hip_get_kernel_timing(ms, ln)
print("decompress %.3f ms per GiB (%s)" % (ms[2] / 5, os.environ.get("CSNAPPY_HIP_DEC_WGS_PER_CU", "default")))
