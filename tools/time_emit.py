import os, sys
sys.path.insert(0, os.getcwd())
import torch
from csnappy_amd import api
nb = 16384
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
d_in = api.generate(cfg, 0xC5A90001, 0, nb, 65536)
b = api.Batch([65536] * nb)
d_out = torch.zeros(b.out_bytes, dtype=torch.uint8, device="cuda")
L = api.lib()
import ctypes as C
for it in range(3):
    api.compress_batch(d_in, b.d_in_off, b.d_in_len, b.max_in_len, d_out, b.d_out_off, b.d_out_len, 16, 0, b.d_ws)
torch.cuda.synchronize()
L.csnappy_hip_set_kernel_timing(1)
for it in range(5):
    api.compress_batch(d_in, b.d_in_off, b.d_in_len, b.max_in_len, d_out, b.d_out_off, b.d_out_len, 16, 0, b.d_ws)
torch.cuda.synchronize()
ms = (C.c_float * 4)(); ln = (C.c_uint32 * 4)()
L.csnappy_hip_get_kernel_timing(ms, ln)
print(os.environ.get("CSNAPPY_AMD_LIB", "default").split("/")[-2] if os.environ.get("CSNAPPY_AMD_LIB") else "default", "parse %.3f emit %.3f ms per GiB" % (ms[0] / 5, ms[1] / 5))
