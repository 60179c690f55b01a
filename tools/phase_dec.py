#!/usr/bin/env python3
"""Development: s_memtime phase counters of the decompress kernel (run on the GPU box against a
library built with tools/build_variant.sh <name> -DCSNAPPY_DEC_PROF=1 [...]).
usage: CSNAPPY_AMD_LIB=build/var/<name>/libcsnappy.so phase_dec.py [text|low|page|urls]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from csnappy_amd import api

kind, seed, block, p, mode = {"text": (0, 0xC5A90001, 65536, 16, 0), "low": (1, 0xC5A90005, 65536, 16, 0),
                              "page": (2, 0xC5A90004, 4096, 13, 1), "urls": (-1, 0, 65536, 16, 0)}[sys.argv[1] if len(sys.argv) > 1 else "text"]
nb = (int(os.environ.get('PHASE_DEC_MIB', '1024')) << 20) // block
if kind >= 0:
    d_in = api.generate(kind, seed, 0, nb, block)
else:
    raw = np.fromfile(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "urls.10K"),
                      dtype=np.uint8)
    d_in = torch.from_numpy(np.resize(raw, nb * block)).cuda()
b = api.Batch([block] * nb)
d_out = torch.zeros(b.out_bytes, dtype=torch.uint8, device="cuda")
api.compress_batch(d_in, b.d_in_off, b.d_in_len, b.max_in_len, d_out, b.d_out_off, b.d_out_len, p, mode, b.d_ws)
d_back = torch.zeros(nb * block, dtype=torch.uint8, device="cuda")
back_off = torch.arange(nb, dtype=torch.int64, device="cuda") * block
cap = torch.full((nb,), block, dtype=torch.int32, device="cuda")
status = torch.zeros(nb, dtype=torch.int32, device="cuda")
produced = torch.zeros(nb, dtype=torch.int32, device="cuda")
L = api.lib()
L.csnappy_hip_debug_dec_prof.argtypes = [C.c_void_p]
L.csnappy_hip_debug_dec_prof.restype = C.c_int
buf = (C.c_ulonglong * 32)()
size_mib = int(os.environ.get("PHASE_DEC_MIB", "1024"))
for it in range(2):
    torch.cuda.synchronize()
    L.csnappy_hip_debug_dec_prof(buf)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    api.decompress_batch(d_out, b.d_out_off, b.d_out_len, d_back, back_off, cap, status, produced, mode)
    t1.record()
    torch.cuda.synchronize()
assert L.csnappy_hip_debug_dec_prof(buf) == 0
assert torch.equal(d_back, d_in) and int(status.abs().sum()) == 0
v = list(buf)
nw, nbat, nscan = max(v[24], 1), max(v[16], 1), max(v[17], 1)
print(f"{sys.argv[1] if len(sys.argv) > 1 else 'text'}: kernel {t0.elapsed_time(t1):.3f} ms for {nb * block >> 20} MiB; waves {nw}")
print(f"  per wave: {v[0]/nw:10.0f} ticks;  batches {v[16]/nw:.1f}  scan iterations/batch {v[17]/nbat:.2f}  "
      f"elements/batch {v[18]/nbat:.1f}  dependent copies/batch {v[19]/nbat:.2f}  long literals/batch {v[20]/nbat:.3f}  "
      f"direct literals/wave {v[21]/nw:.2f}")
names = ["scan: wait for the bytes", "scan: decode", "scan: walk", "scan: request + queue write", "queue read, checks, offsets",
         "-", "one-lane copies (round trip)", "literals > 64", "dependent copies", "flush",
         "direct long literal", "other"]
for i, n in enumerate(names):
    print(f"  {n:32s} {v[1+i]/nbat:9.1f} ticks per batch  {100.0*v[1+i]/max(v[0],1):5.1f} %")
