#!/usr/bin/env python3
"""Why did csnappy_hip_decompress_stream leave the fragments (GPU box)?  Decodes one stream with
the stream call and prints the index's flags, totals and the first missing fragment boundaries.
usage: stream_debug.py FILE.snappy   |   stream_debug.py --synthetic NBYTES"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from csnappy_amd import api


def layout(n, ulen):
    nseg, nfrag = (n + 4095) // 4096, (ulen + 32767) // 32768
    at, out = 0, {}
    for name, size in (("tagmask", nseg * 512), ("winout", nseg * 256), ("truemask", nseg * 512), ("trueout", nseg * 256),
                       ("seg_out", nseg * 8), ("seg_exit", nseg * 4), ("seg_xesz", nseg * 4), ("seg_safe", nseg * 4),
                       ("seg_leave", nseg * 4), ("last_tag", nseg * 8192), ("seg_entry", nseg * 4),
                       ("grp_e", ((nseg + 63) // 64) * 4), ("grp_esz", ((nseg + 63) // 64) * 4),
                       ("f_in_off", nfrag * 8), ("f_out_off", nfrag * 8), ("frag_pos", nfrag * 4),
                       ("f_in_len", nfrag * 4), ("f_out_cap", nfrag * 4), ("f_produced", nfrag * 4), ("f_status", nfrag * 4),
                       ("one_off", 16), ("total", 8), ("one_len", 8), ("flags", 32)):
        out[name] = (at, size)
        at += (size + 15) & ~15
    return out, at, nseg, nfrag


def main():
    if sys.argv[1] == "--synthetic":
        n = int(sys.argv[2])
        data = api.generate_host(api.WG_TEXT, 1, 0, 1, n).tobytes()
        stream = api.compress(data, 16)
    else:
        stream = open(sys.argv[1], "rb").read()
    hdr, ulen = 0, 0
    while True:
        c = stream[hdr]
        ulen |= (c & 127) << (7 * hdr)
        hdr += 1
        if c < 128:
            break
    body = torch.from_numpy(np.frombuffer(stream[hdr:], dtype=np.uint8).copy()).cuda()
    n = body.numel()
    lay, total, nseg, nfrag = layout(n, ulen)
    L = api.lib()
    assert L.csnappy_hip_decompress_stream_workspace_size(n, ulen) == total, "layout out of date"
    ws = torch.zeros(total + 16, dtype=torch.uint8, device="cuda")
    ws = ws[(-ws.data_ptr()) % 16:]
    res = torch.zeros(2, dtype=torch.int32, device="cuda")
    d_out = torch.zeros(ulen + 64, dtype=torch.uint8, device="cuda")
    rc = L.csnappy_hip_decompress_stream(body.data_ptr(), n, ulen, d_out.data_ptr(), res.data_ptr(), res.data_ptr() + 4,
                                         ws.data_ptr(), total, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    h = ws.cpu().numpy()
    get = lambda name, dt: h[lay[name][0]:lay[name][0] + lay[name][1]].view(dt)
    flags = get("flags", np.uint32)
    print(f"rc {rc} status/produced {res.cpu().tolist()}  body {n} B, {nseg} segments; expects {ulen} B, {nfrag} fragments")
    print(f"refused bits {flags[0]} (1 chain and settle disagree, 2 huge element, 4 missing boundary)  parse ends at {flags[1]} "
          f"(body {n})  verdict {flags[2]}  grain {flags[3]}  total out {get('total', np.uint64)[0]}")
    ent, safe = get("seg_entry", np.uint32), get("seg_safe", np.uint32)
    inside = ent != 0xFFFFFFFF
    rel = (ent.astype(np.int64) - np.arange(nseg) * 4096)[inside]
    print(f"segments entered {int(inside.sum())} of {nseg}; entries beyond the safe prefix (table path or long element): "
          f"{int((rel >= safe[inside]).sum())}; median safe prefix {int(np.median(safe))} B")
    # time: the stream call against the same stream on one wave (the batch call with one block)
    def timed(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps
    st = torch.cuda.current_stream().cuda_stream
    t_stream = timed(lambda: L.csnappy_hip_decompress_stream(body.data_ptr(), n, ulen, d_out.data_ptr(), res.data_ptr(),
                                                              res.data_ptr() + 4, ws.data_ptr(), total, st))
    api.set_kernel_timing(True)
    api.get_kernel_timing()
    L.csnappy_hip_decompress_stream(body.data_ptr(), n, ulen, d_out.data_ptr(), res.data_ptr(), res.data_ptr() + 4,
                                    ws.data_ptr(), total, st)
    torch.cuda.synchronize()
    ms, cnt = (api.C.c_float * 4)(), (api.C.c_uint32 * 4)()
    L.csnappy_hip_get_kernel_timing(ms, cnt)
    api.set_kernel_timing(False)
    print(f"  of which index kernels {ms[3]:.3f} ms, fragment decode + verdict + (skipped) one-wave launch {ms[2]:.3f} ms")
    whole = torch.from_numpy(np.frombuffer(stream, dtype=np.uint8).copy()).cuda()
    zero = torch.zeros(1, dtype=torch.int64, device="cuda")
    ilen = torch.tensor([len(stream)], dtype=torch.int32, device="cuda")
    cap = torch.tensor([ulen], dtype=torch.int32, device="cuda")
    r2 = torch.zeros(2, dtype=torch.int32, device="cuda")
    t_one = timed(lambda: api.decompress_batch(whole, zero, ilen, d_out, zero, cap, r2[:1], r2[1:], api.STREAM), reps=2)
    print(f"stream call {t_stream:.3f} ms ({ulen / t_stream / 1e6:.2f} GB/s of output)   one wave {t_one:.3f} ms "
          f"({ulen / t_one / 1e6:.2f} GB/s)   x{t_one / t_stream:.1f}")
    pos = get("frag_pos", np.uint32)
    missing = np.flatnonzero(pos == 0xFFFFFFFF)
    print(f"boundaries missing: {len(missing)} of {nfrag}; first {missing[:10].tolist()}")
    st, pr, cap = get("f_status", np.int32), get("f_produced", np.uint32), get("f_out_cap", np.uint32)
    badf = np.flatnonzero((st != 0) | (pr != cap))
    print(f"fragments not clean: {len(badf)}; first {[(int(f), int(st[f]), int(pr[f]), int(cap[f])) for f in badf[:8]]}")
    ent = get("seg_entry", np.uint32)
    print("entries", ent[:8].tolist(), "exits", get("seg_exit", np.uint32)[:8].tolist(), "seg offsets", get("seg_out", np.uint64)[:8].tolist())


main()
