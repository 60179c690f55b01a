#!/usr/bin/env python3
"""Copy a profile produced by tools/profile_gpu.sh from gpurun_out/ (scratch) into profiles/
(tracked) and record the measured HBM traffic of the dominant kernel for bench.py.

    python tools/profile_commit.py <tag> <workload>_p<power>      e.g.  r01_text_p16 text_p16
"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, key = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "summary.json"), os.path.join(dst, f"{tag}_summary.json"))
shutil.copy(os.path.join(src, "stats", "run_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
s = json.load(open(os.path.join(src, "summary.json")))
path = os.path.join(dst, "pmc_traffic.json")
t = json.load(open(path)) if os.path.exists(path) else {}
e = {"source": f"profiles/{tag}_summary.json (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes; "
               "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024, gfx950 wide-read correction)"}
e["note"] = ("per batch call (all chunk launches of the kernel summed); FETCH_SIZE is doubled as the guide prescribes for "
             "gfx950, which is calibrated for wide coalesced reads only -- the raw counters are in the summary")
for k, v in s["per_batch"].items():
    if isinstance(v, dict) and "hbm_bytes_corrected" in v:
        e[f"{k}_bytes_per_batch"] = v["hbm_bytes_corrected"]
t[key] = e
json.dump(t, open(path, "w"), indent=1, sort_keys=True)
print("committed", tag, "->", dst, e)
