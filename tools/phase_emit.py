#!/usr/bin/env python3
"""Development: s_memtime phase counters of the emit kernel's per-chunk work (run on the GPU box
against a library built with tools/build_variant.sh <name> -DCSNAPPY_EMIT_PROF=1).
usage: CSNAPPY_AMD_LIB=build/var/<name>/libcsnappy.so [PHASE_MIB=1024] phase_emit.py [text|low|urls]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from csnappy_amd import api

w = sys.argv[1] if len(sys.argv) > 1 else "text"
kind, seed, block, p, mode = {"text": (0, 0xC5A90001, 65536, 16, 0), "low": (1, 0xC5A90005, 65536, 16, 0),
                              "urls": (-1, 0, 65536, 16, 0)}[w]
nb = (int(os.environ.get("PHASE_MIB", "1024")) << 20) // block
if kind >= 0:
    d_in = api.generate(kind, seed, 0, nb, block)
else:
    raw = np.fromfile(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "urls.10K"),
                      dtype=np.uint8)
    d_in = torch.from_numpy(np.resize(raw, nb * block)).cuda()
b = api.Batch([block] * nb)
d_out = torch.zeros(b.out_bytes, dtype=torch.uint8, device="cuda")
L = api.lib()
L.csnappy_hip_debug_emit_prof.argtypes = [C.c_void_p]
L.csnappy_hip_debug_emit_prof.restype = C.c_int
buf = (C.c_ulonglong * 16)()
for it in range(2):
    torch.cuda.synchronize()
    L.csnappy_hip_debug_emit_prof(buf)
    api.compress_batch(d_in, b.d_in_off, b.d_in_len, b.max_in_len, d_out, b.d_out_off, b.d_out_len, p, mode, b.d_ws)
    torch.cuda.synchronize()
assert L.csnappy_hip_debug_emit_prof(buf) == 0
v = list(buf)
n = max(v[8], 1)
print(f"{w}: {nb * block >> 20} MiB, {n} chunks of 64 records, {v[9]} waves; {v[0]/n:.0f} cycles per chunk (wave lifetime / its chunks)")
for i, name in enumerate(["decode, offsets, wait for the chunk's loads", "literal payload into the staging", "literal headers, copy tags",
                          "big records", "drain"]):
    print(f"  {name:46s} {v[1+i]/n:9.1f} cycles per chunk  {100.0*v[1+i]/max(v[0],1):5.1f} %")
