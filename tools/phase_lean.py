#!/usr/bin/env python3
"""Debug: s_memtime phase counters of the lean compress parser (run on the GPU box).
usage: phase_lean.py [text|low|page|urls] [p]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from csnappy_amd import api

kind, seed, block, p, mode = {"text": (0, 0xC5A90001, 65536, 16, 0), "low": (1, 0xC5A90005, 65536, 16, 0),
                              "page": (2, 0xC5A90004, 4096, 13, 1), "urls": (-1, 0, 65536, 16, 0)}[sys.argv[1] if len(sys.argv) > 1 else "text"]
if len(sys.argv) > 2:
    p = int(sys.argv[2])
nb = (int(os.environ.get('PHASE_MIB', '256')) << 20) // block
if kind >= 0:
    d_in = api.generate(kind, seed, 0, nb, block)
else:
    raw = np.fromfile(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "urls.10K"),
                      dtype=np.uint8)
    d_in = torch.from_numpy(np.resize(raw, nb * block)).cuda()
b = api.Batch([block] * nb)
d_out = torch.zeros(b.out_bytes, dtype=torch.uint8, device="cuda")
prof = torch.zeros(24, dtype=torch.int64, device="cuda")
L = api.lib()
L.csnappy_hip_debug_set_profile_buffer.argtypes = [C.c_void_p]
for it in range(2):
    prof.zero_()
    L.csnappy_hip_debug_set_profile_buffer(prof.data_ptr())
    api.compress_batch(d_in, b.d_in_off, b.d_in_len, b.max_in_len, d_out, b.d_out_off, b.d_out_len, p, mode, b.d_ws)
    torch.cuda.synchronize()
L.csnappy_hip_debug_set_profile_buffer(None)
v = prof.cpu().tolist()
nf, steps = max(v[14], 1), max(v[10], 1)
print(f"fragments {nf}  steps/frag {v[10]/nf:.1f}  copies/frag {v[11]/nf:.1f}  special visits/frag {v[12]/nf:.1f}  sparse steps/frag {v[13]/nf:.1f}")
print(f"  parser total {v[0]/nf:12.0f} ticks/frag {v[0]/steps:9.1f} per step;  prologue {v[15]/nf:10.0f} ticks/frag")
print(f"  per dense step: tabbed lanes {v[16]/steps:.1f}  gathered {v[17]/steps:.1f}  4-byte matches {v[18]/steps:.1f}  flagged {v[19]/steps:.1f}")
names = ["commit (rest of prev step)", "wait own bytes+ids", "filters + table", "candidate gather", "match len + next-stop",
         "chain walk", "cursor + place", "records", "wait next bytes"]
for i, n in enumerate(names):
    print(f"  {n:28s} {v[1+i]/nf:12.0f} ticks/frag {v[1+i]/steps:9.1f} per step")
