#!/usr/bin/env python3
"""Development: throughput of the framing-format calls (host buffers) on one large buffer."""
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
cap0 = L.csnappy_frame_max_compressed_length(len(x))
out = np.empty(cap0 + 8, dtype=np.uint8)
back = np.empty(len(x), dtype=np.uint8)
api.frame_compress(x[:1 << 20])
for it in range(3):
    cap = C.c_size_t(cap0)
    t0 = time.perf_counter()
    rc = L.csnappy_frame_compress(x.ctypes.data, len(x), out.ctypes.data, C.byref(cap), 16)
    t1 = time.perf_counter()
    n = C.c_size_t(len(x))
    rc2 = L.csnappy_frame_decompress(out.ctypes.data, cap.value, back.ctypes.data, C.byref(n))
    t2 = time.perf_counter()
    assert rc == 0 and rc2 == 0 and n.value == len(x) and np.array_equal(back, x)
    print(f"csnappy_frame_compress {mib / 1024 / (t1 - t0):.2f} GiB/s  csnappy_frame_decompress {mib / 1024 / (t2 - t1):.2f} GiB/s "
          f"(host to host, {cap.value} framed bytes)")
