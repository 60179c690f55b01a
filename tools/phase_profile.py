#!/usr/bin/env python3
"""Debug: s_memtime phase counters of the compress parser (run on the GPU box)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from csnappy_amd import api

kind, seed, block, p, mode = {"text": (0, 0xC5A90001, 65536, 16, 0), "low": (1, 0xC5A90005, 65536, 16, 0),
                              "page": (2, 0xC5A90004, 4096, 13, 1), "urls": (-1, 0, 65536, 16, 0)}[sys.argv[1] if len(sys.argv) > 1 else "text"]
if len(sys.argv) > 2:
    p = int(sys.argv[2])
nb = (256 << 20) // block
if kind >= 0:
    d_in = api.generate(kind, seed, 0, nb, block)
else:
    import numpy as np
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
nf = v[9]
names = ["total", "vec", "walk", "commit", "publish"]
print("table mode:", os.environ.get("CSNAPPY_HIP_TABLE", "auto (by LDS occupancy)"))
print(f"fragments {nf}  steps/frag {v[5]/nf:.1f}  matches/frag {v[6]/nf:.1f}  wide/frag {v[7]/nf:.1f}  sparse/frag {v[8]/nf:.1f}")
for i, n in enumerate(names):
    print(f"  {n:8s} {v[i]/nf:12.0f} ticks/frag   {v[i]/max(v[5],1):9.1f} per step   {v[i]/max(v[6],1):9.1f} per match")
for i, n in ((10, "w.chain"), (11, "w.stop"), (12, "w.place"), (13, "w.records"), (14, "pre-loop"), (15, "post-loop"), (16, "pro.count"), (17, "pro.prefix"), (18, "pro.ids"), (19, "pro.fence"), (20, "n.hops"), (21, "n.flag_visits"), (22, "n.flag_forwarded"), (23, "n.flagged_lanes")):
    print(f"  {n:9s} {v[i]/nf:11.0f} ticks/frag   {v[i]/max(v[5],1):9.1f} per step")
