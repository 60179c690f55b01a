"""Development: compress of one half-batch and decompress of another, one after the other on one stream and side by side
on two (what pipelining the chunks of a step would give)."""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch
from csnappy_amd import api

nb = 8192
blk = 65536
kind = int(sys.argv[1]) if len(sys.argv) > 1 else 0


class Half:
    def __init__(self, first):
        self.d_in = api.generate(kind, 0xC5A90001, first, nb, blk)
        self.b = api.Batch([blk] * nb)
        self.d_out = torch.zeros(self.b.out_bytes, dtype=torch.uint8, device="cuda")
        self.d_back = torch.zeros(nb * blk, dtype=torch.uint8, device="cuda")
        self.cap = torch.full((nb,), blk, dtype=torch.int32, device="cuda")
        self.boff = torch.arange(nb, dtype=torch.int64, device="cuda") * blk
        self.status = torch.zeros(nb, dtype=torch.int32, device="cuda")
        self.prod = torch.zeros(nb, dtype=torch.int32, device="cuda")

    def compress(self):
        b = self.b
        api.compress_batch(self.d_in, b.d_in_off, b.d_in_len, b.max_in_len, self.d_out, b.d_out_off, b.d_out_len, 16, 0, b.d_ws)

    def decompress(self):
        b = self.b
        api.decompress_batch(self.d_out, b.d_out_off, b.d_out_len, self.d_back, self.boff, self.cap, self.status, self.prod, 0)


A, B = Half(0), Half(nb)
for h in (A, B):
    h.compress()
    h.decompress()
torch.cuda.synchronize()
assert torch.equal(B.d_back, B.d_in[: nb * blk])
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, reps=5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def serial():
    A.compress()
    B.decompress()


def side_by_side():
    with torch.cuda.stream(s1):
        A.compress()
    with torch.cuda.stream(s2):
        B.decompress()


def only_c():
    A.compress()


def only_d():
    B.decompress()


for name, fn in (("compress alone", only_c), ("decompress alone", only_d), ("one after the other", serial), ("side by side", side_by_side)):
    fn()
    print("%-22s %.3f ms for 0.5 GiB each" % (name, timed(fn)))
torch.cuda.synchronize()
assert torch.equal(B.d_back, B.d_in[: nb * blk])
A.decompress()
torch.cuda.synchronize()
assert torch.equal(A.d_back, A.d_in[: nb * blk])
print("results equal")
