"""Development: the dense step loop's phases by s_memtime (a build with -DCSNAPPY_ISA_PROF=1:
tools/build_variant.sh isaprof -DCSNAPPY_ISA_PROF=1; CSNAPPY_AMD_LIB=build/var/isaprof/libcsnappy.so python tools/phase_isa.py)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.getcwd())
import torch
from csnappy_amd import api

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
nb = 16384
d_in = api.generate(cfg, 0xC5A90001, 0, nb, 65536)
b = api.Batch([65536] * nb)
d_out = torch.zeros(b.out_bytes, dtype=torch.uint8, device="cuda")
L = api.lib()
buf = (C.c_ulonglong * 16)()
for it in range(2):
    api.compress_batch(d_in, b.d_in_off, b.d_in_len, b.max_in_len, d_out, b.d_out_off, b.d_out_len, 16, 0, b.d_ws)
torch.cuda.synchronize()
L.csnappy_hip_debug_isa_prof.argtypes = [C.c_void_p]
assert L.csnappy_hip_debug_isa_prof(buf) == 0
api.compress_batch(d_in, b.d_in_off, b.d_in_len, b.max_in_len, d_out, b.d_out_off, b.d_out_len, 16, 0, b.d_ws)
torch.cuda.synchronize()
assert L.csnappy_hip_debug_isa_prof(buf) == 0
names = ["commit -> front", "table (read + add back)", "gather wait", "compare, masks, next stops", "walk (+ visits)", "back (cursor, loads, records)"]
steps = buf[7]
tot = sum(buf[k] for k in range(6))
print("loop entries %d, steps %d, ticks per step %.1f (s_memtime: 100 MHz)" % (buf[6], steps, tot / max(steps, 1)))
for k in range(6):
    print("  %-32s %6.2f ticks per step  %5.1f %%" % (names[k], buf[k] / max(steps, 1), 100.0 * buf[k] / max(tot, 1)))
