import os, sys
sys.path.insert(0, os.getcwd())
import torch, ctypes as C
from csnappy_amd import api
n = 256 << 20
d_in = api.generate(0, 0xC5A90001, 0, n // 65536, 65536)
b = api.Batch([n])
d_out = torch.zeros(b.out_bytes, dtype=torch.uint8, device="cuda")
api.lib().csnappy_hip_set_kernel_timing(1)
for it in range(3):
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    api.compress_batch(d_in, b.d_in_off, b.d_in_len, b.max_in_len, d_out, b.d_out_off, b.d_out_len, 16, 0, b.d_ws)
    t1.record(); torch.cuda.synchronize()
    print("one 256 MiB stream: compress %.2f ms -> %.1f GiB/s" % (t0.elapsed_time(t1), 0.25 / (t0.elapsed_time(t1) / 1e3)))
ms = (C.c_float * 4)(); ln = (C.c_uint32 * 4)()
api.lib().csnappy_hip_get_kernel_timing(ms, ln)
print("parse ms", ms[0] / max(ln[0], 1), "emit ms", ms[1] / max(ln[1], 1), "out_len", int(b.d_out_len[0]))

# ---- the same stream back through the stream call (tag index + one wave per fragment) ----
body_off = 0
c = d_out[: int(b.d_out_len[0])]
hdr = bytes(c[:5].cpu().numpy())
ulen, k, shift = 0, 0, 0
while True:
    ulen |= (hdr[k] & 0x7F) << shift
    k += 1
    if hdr[k - 1] < 128:
        break
    shift += 7
d_back = torch.empty(n, dtype=torch.uint8, device="cuda")
for it in range(3):
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    st, produced, fast = api.decompress_stream(c[k:], ulen, d_back)
    t1.record(); torch.cuda.synchronize()
    print("one 256 MiB stream: decompress %.2f ms -> %.1f GiB/s (status %d, fragment path %s)"
          % (t0.elapsed_time(t1), 0.25 / (t0.elapsed_time(t1) / 1e3), st, fast))
assert torch.equal(d_back, d_in.view(-1)[:n])
