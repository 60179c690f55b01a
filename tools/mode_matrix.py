#!/usr/bin/env python3
"""Compress/decompress kernel time per workload x table mode x table power (run on the GPU box).
usage: mode_matrix.py [workload:p,p,..]...   (default: the round-2 sweep)"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = sys.argv[1:] or ["text:16,15,14,13,12", "urls:16,14", "low:16", "page:13"]
for item in spec:
    wl, ps = item.split(":")
    for p in (int(x) for x in ps.split(",")):
        for mode in ("hash", "dense", "global"):
            env = dict(os.environ, CSNAPPY_HIP_TABLE=mode)
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1",
                                  "--no-cpu-baseline", "--workload", wl, "--p", str(p), "--gib", "0.5"],
                                 env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
            try:
                d = json.loads(out.strip().splitlines()[-1])
                print(f"{wl:5s} p={p:2d} {mode:6s} compress {d['compress_gibs']:8.2f} GiB/s  decompress {d['decompress_gibs']:8.2f} GiB/s  "
                      f"ratio {d['compressed_ratio']:.3f}  round-trip {d['value']:.2f}", flush=True)
            except Exception as e:
                print(wl, p, mode, "FAILED", e, out[-300:], flush=True)
