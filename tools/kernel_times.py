#!/usr/bin/env python3
"""Per-kernel ms of one bench.py run (GPU box): usage kernel_times.py [bench.py args...]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-other-configs"] + sys.argv[1:],
                     stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
d = json.loads(out.strip().splitlines()[-1])
print(" ".join(sys.argv[1:]), "| compress %.2f decompress %.2f GiB/s |" % (d["compress_gibs"], d["decompress_gibs"]),
      " ".join(f"{k.replace('snappy_', '')}={v['ms_per_step']:.3f}ms" for k, v in d["kernels"].items()), flush=True)
