#!/usr/bin/env python3
"""Development (GPU box): compress a generated workload on the GPU and print the first blocks whose bytes differ
from the CPU checker's, with the first differing element of each.   usage: tests/dbg_blocks.py [text|low|page] [MiB]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, oracle
from csnappy_amd import api
w = sys.argv[1] if len(sys.argv) > 1 else "low"
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 64
kind, seed, block, p, mode = {"text": (0, 0xC5A90001, 65536, 16, 0), "low": (1, 0xC5A90005, 65536, 16, 0),
                              "page": (2, 0xC5A90004, 4096, 13, 1)}[w]
nb = (mib << 20) // block
d_in = api.generate(kind, seed, 0, nb, block)
b = api.Batch([block] * nb)
d_out = torch.zeros(b.out_bytes, dtype=torch.uint8, device="cuda")
api.compress_batch(d_in, b.d_in_off, b.d_in_len, b.max_in_len, d_out, b.d_out_off, b.d_out_len, p, mode, b.d_ws)
torch.cuda.synchronize()
host = d_in.cpu().numpy(); out = d_out.cpu().numpy(); lens = b.d_out_len.cpu().numpy()
P = oracle.best()
def elements(s, skip_header):
    i = 0
    if skip_header:
        while s[i] & 0x80: i += 1
        i += 1
    pos = 0; els = []
    while i < len(s):
        t = s[i]; k = t & 3
        if k == 0:
            n = t >> 2
            if n < 60: n += 1; i += 1
            else:
                nb_ = n - 59; n = int.from_bytes(bytes(s[i+1:i+1+nb_]), "little") + 1; i += 1 + nb_
            els.append(("lit", pos, n)); i += n; pos += n
        elif k == 1:
            n = 4 + ((t >> 2) & 7); off = ((t >> 5) << 8) | s[i+1]; i += 2; els.append(("copy", pos, n, off)); pos += n
        elif k == 2:
            n = (t >> 2) + 1; off = s[i+1] | (s[i+2] << 8); i += 3; els.append(("copy", pos, n, off)); pos += n
        else:
            n = (t >> 2) + 1; off = int.from_bytes(bytes(s[i+1:i+5]), "little"); i += 5; els.append(("copy", pos, n, off)); pos += n
    return els
bad = 0
for i in range(nb):
    x = host[i * block:(i + 1) * block]
    want = P.compress(x, p) if mode == 0 else P.compress_fragment(x, p)
    o = int(b.out_off[i]); got = bytes(out[o:o + int(lens[i])])
    if got != want:
        bad += 1
        if bad <= 4:
            eg, ew = elements(got, mode == 0), elements(want, mode == 0)
            k = next((j for j in range(min(len(eg), len(ew))) if eg[j] != ew[j]), min(len(eg), len(ew)))
            print(f"block {i}: len got {len(got)} want {len(want)}; first differing element #{k}")
            print("  got ", eg[max(0, k - 2):k + 3]); print("  want", ew[max(0, k - 2):k + 3])
            q = ew[k][1] if k < len(ew) else 0
            print("  input around", q, ":", bytes(x[max(0, q - 24):q + 40]).hex())
print("blocks", nb, "bad", bad)
bl = [i for i in range(nb) if bytes(out[int(b.out_off[i]):int(b.out_off[i]) + int(lens[i])]) != (P.compress(host[i * block:(i + 1) * block], p) if mode == 0 else P.compress_fragment(host[i * block:(i + 1) * block], p))]
rng = []
for i in bl:
    if rng and rng[-1][1] == i - 1: rng[-1][1] = i
    else: rng.append([i, i])
print("bad ranges:", rng[:40])
