#!/usr/bin/env python3
"""Development (GPU box): the parser's records of one fragment, from two libraries, side by side.
usage: tests/dbg_records.py <libA> <libB> [workload] [MiB] [fragment]   (a library path, or 'default')"""
import os, sys, subprocess, json
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np, torch
    from csnappy_amd import api
    w, mib, frag = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    kind, seed, block, p, mode = {"text": (0, 0xC5A90001, 65536, 16, 0), "low": (1, 0xC5A90005, 65536, 16, 0)}[w]
    nb = (mib << 20) // block
    d_in = api.generate(kind, seed, 0, nb, block)
    b = api.Batch([block] * nb)
    d_out = torch.zeros(b.out_bytes, dtype=torch.uint8, device="cuda")
    api.compress_batch(d_in, b.d_in_off, b.d_in_len, b.max_in_len, d_out, b.d_out_off, b.d_out_len, p, mode, b.d_ws)
    torch.cuda.synchronize()
    ws = b.d_ws.cpu().numpy()
    nfr = nb * 2
    cnt_bytes = (nfr * 4 + 255) & ~255
    rec_cap = 32768 // 4 + 8
    cnt = ws[:nfr * 4].view(np.uint32)
    recs = ws[cnt_bytes:cnt_bytes + nfr * rec_cap * 8].view(np.uint32).reshape(nfr, rec_cap, 2)
    n = int(cnt[frag])
    out = [[int(r[0] & 0xffff), int(r[0] >> 16), int(r[1] & 0xffff), int(r[1] >> 16)] for r in recs[frag][:n]]
    print(json.dumps(out))
    sys.exit(0)
la, lb = sys.argv[1], sys.argv[2]
w = sys.argv[3] if len(sys.argv) > 3 else "low"
mib = sys.argv[4] if len(sys.argv) > 4 else "64"
frag = sys.argv[5] if len(sys.argv) > 5 else "1024"
def run(lib):
    env = dict(os.environ)
    if lib != "default": env["CSNAPPY_AMD_LIB"] = os.path.abspath(lib)
    o = subprocess.run([sys.executable, __file__, "--child", w, mib, frag], env=env, capture_output=True, text=True)
    return json.loads(o.stdout.strip().splitlines()[-1])
a, b = run(la), run(lb)
print("records (base, cand, len, lit_start):", len(a), len(b))
k = next((i for i in range(min(len(a), len(b))) if a[i] != b[i]), None)
print("first difference at record", k)
if k is not None:
    for i in range(max(0, k - 3), min(len(a), k + 4)):
        print(i, a[i], b[i] if i < len(b) else None)
