#!/usr/bin/env python3
"""Development: throughput of the LEGACY calls (host buffers: two PCIe copies + one launch per call)
on one large buffer -- the PCIe-inclusive rate DESIGN.md §5 quotes.  usage: legacy_rate.py [MiB]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from csnappy_amd import api

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = np.frombuffer(api.generate_host(0, 0xC5A90001, 0, (mib << 20) // 65536, 65536), dtype=np.uint8)
L = api.lib()
out = np.empty(api.max_compressed_length(len(x)) + 8, dtype=np.uint8)
back = np.empty(len(x), dtype=np.uint8)
n = C.c_uint32(0)
api.compress(x[:1 << 20])  # context, buffers
for name in ("first", "second", "third"):
    t0 = time.perf_counter()
    L.csnappy_compress(x.ctypes.data, len(x), out.ctypes.data, C.byref(n), None, 16)
    t1 = time.perf_counter()
    rc = L.csnappy_decompress(out.ctypes.data, n.value, back.ctypes.data, len(x))
    t2 = time.perf_counter()
    assert rc == 0 and np.array_equal(back, x)
    print(f"{name}: csnappy_compress {mib} MiB host to host {mib / 1024 / (t1 - t0):.2f} GiB/s, "
          f"csnappy_decompress {mib / 1024 / (t2 - t1):.2f} GiB/s (ratio {n.value / len(x):.3f})")
