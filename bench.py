#!/usr/bin/env python3
"""bench.py -- throughput of the Snappy block-codec hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no launcher around it (no RANK in the environment) bench.py starts the N ranks
itself, as a child `python -m torch.distributed.run --standalone --nproc-per-node N bench.py ...`,
and exits with that child's code; under a launcher (RANK / LOCAL_RANK / WORLD_SIZE set) it is one rank.

A "step" is one pass of the hot path over one batch that is already resident in HBM:
compress every block of the batch (snappy_parse_fragments + the snappy_emit_* launches), then
decompress every block back (snappy_decompress_blocks).  The default workload is BASELINE.json
configs[1]: 1 GiB of the G_text synthetic per GPU, cut into 65536-byte blocks, STREAM mode
(csnappy_compress / csnappy_decompress semantics), table power 16.  Blocks are independent, so
N GPUs take N disjoint block ranges (weak scaling: 1 GiB per rank), with no data-path collective.

Rank 0 prints ONE JSON line.  `value` = uncompressed bytes round-tripped per second, whole job.
`roofline` is for the dominant kernel, from HIP events recorded on the launch stream inside the
timed region.  `cpu_baseline` is the reference C code (oracle/_ref, "reference") or its
restatement (oracle/, "port") on this box's host cores over a bounded sample of the same input.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md

WORKLOADS = {
    # name: (generator kind, seed, block bytes, table power, mode, description)
    "text": (0, 0xC5A90001, 65536, 16, 0, "G_text synthetic (URL-like tokens)"),
    "low": (1, 0xC5A90005, 65536, 16, 0, "G_low synthetic (runs / short periods)"),
    "page": (2, 0xC5A90004, 4096, 13, 1, "G_page synthetic (zram-style 4 KiB page mix)"),
    "urls": (-1, 0, 65536, 16, 0, "testdata/urls.10K replicated end to end"),
}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def load_measured_traffic(workload, p, kernels):
    """HBM bytes per batch call of `kernels` (summed) from committed rocprofv3 PMC passes
    (profiles/pmc_traffic.json, written by tools/profile_commit.py), or None."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        t = json.load(open(path)).get(f"{workload}_p{p}", {})
        vals = [t.get(f"{k}_bytes_per_batch") for k in kernels]
        return None if any(v is None for v in vals) else int(sum(vals))
    except (OSError, ValueError):
        return None


def measured_copy_bandwidth(torch, nbytes):
    """Streaming-copy bandwidth of this box: device-to-device copy of an nbytes
    buffer, bytes read + bytes written per second, best of 5 (GB/s)."""
    a = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    b_ = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    a.zero_()
    best = 0.0
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b_.copy_(a)
        e1.record()
        e1.synchronize()
        best = max(best, 2.0 * nbytes / (e0.elapsed_time(e1) / 1e3) / 1e9)
    del a, b_
    return round(best, 1)


def cpu_baseline(kind, seed, block, p, mode, nblocks_avail, target_s, urls=None):
    """Time the CPU path on a bounded sample of the same workload, all host cores."""
    import oracle
    from csnappy_amd import api
    codec = oracle.best()
    cores = os.cpu_count() or 1

    def sample(nb):
        if kind >= 0:
            return api.generate_host(kind, seed, 0, nb, block)
        rep = np.frombuffer(urls, dtype=np.uint8)
        return np.resize(rep, nb * block)

    bufs = {}

    def run(nb, threads):
        # buffers are allocated and touched once, outside the timed calls
        if bufs.get("nb") != nb:
            b = api.Batch([block] * nb, device=None)
            bufs.update(nb=nb, host=sample(nb), b=b, comp=np.ones(b.out_bytes + 64, dtype=np.uint8),
                        back=np.ones(nb * block + 64, dtype=np.uint8), cap=np.full(nb, block, dtype=np.uint32))
        host, b = bufs["host"], bufs["b"]
        t0 = time.perf_counter()
        out, out_len = oracle.batch_compress(codec, host, b.in_off, b.in_len, b.out_off, b.out_bytes,
                                             p, mode, threads=threads, out=bufs["comp"])
        t1 = time.perf_counter()
        back, status, _ = oracle.batch_decompress(codec, bufs["comp"], b.out_off, out_len, b.in_off, bufs["cap"],
                                                  nb * block, mode, threads=threads, out=bufs["back"])
        t2 = time.perf_counter()
        assert (status == 0).all() and np.array_equal(back[:nb * block], host)
        return t1 - t0, t2 - t1

    probe = max(cores * 4, (8 << 20) // block)
    tc, td = run(probe, cores)
    per_block = (tc + td) / probe
    nb = int(min(nblocks_avail, max(probe, target_s / per_block)))
    nb = min(nb, (4 << 30) // block)  # bound host memory
    # repeat the sample until ~target_s of wall time has been timed (the probe above is cold --
    # thread start-up, page faults -- so the repetition count is decided on the clock, not on it)
    reps, tc, td = 0, 0.0, 0.0
    while reps < 1 or (tc + td < target_s and reps < 400):
        a, b_ = run(nb, cores)
        tc += a
        td += b_
        reps += 1
    gib = nb * block * reps / 2.0 ** 30
    # the same path on ONE thread as well (a ~2 s sample)
    nb1 = int(max(16, min(nb, 2.0 / max(per_block * cores, 1e-9))))
    run(nb1, 1)
    c1, d1 = run(nb1, 1)
    gib1 = nb1 * block / 2.0 ** 30
    return {
        "value": round(gib / (tc + td), 4), "unit": "GiB/s", "cores": cores, "kind": codec.kind,
        "sample": f"first {nb} blocks x {block} B of the same workload, {reps} repetition(s) "
                  f"({gib:.3f} GiB in total): compress {tc:.2f} s + decompress {td:.2f} s on {cores} "
                  "threads (one block range per thread)",
        "compress_gibs": round(gib / tc, 4), "decompress_gibs": round(gib / td, 4),
        "one_thread": {"value": round(gib1 / (c1 + d1), 4), "compress_gibs": round(gib1 / c1, 4),
                       "decompress_gibs": round(gib1 / d1, 4), "sample": f"first {nb1} blocks, 1 thread"},
    }


def verify_against_reference(torch, api, d_in, b, d_out, chunk, block, p, mode, nv):
    """Compress the first nv blocks of `chunk` on the GPU and with the CPU checker (all host
    cores); compare every length and the sha256 of the compacted streams."""
    import hashlib
    import oracle
    lo, cnt = chunk
    src = d_in[lo * block:(lo + cnt) * block]
    api.compress_batch(src, b.d_in_off[:cnt], b.d_in_len[:cnt], b.max_in_len, d_out, b.d_out_off[:cnt],
                       b.d_out_len[:cnt], p, mode, b.d_ws)
    torch.cuda.synchronize()
    codec = oracle.best()
    host = src[:nv * block].cpu().numpy()
    want, want_len = oracle.batch_compress(codec, host, b.in_off[:nv], b.in_len[:nv], b.out_off[:nv],
                                           int(b.out_off[nv - 1] + b.slot[nv - 1]), p, mode, threads=os.cpu_count() or 1)
    got_len = b.d_out_len[:nv].cpu().numpy().astype(np.uint32)
    got = d_out[:int(b.out_off[nv - 1] + b.slot[nv - 1])].cpu().numpy()
    slot = int(b.slot[0])

    def compact(buf, lens):
        m = np.arange(slot, dtype=np.uint32)[None, :] < lens[:, None]
        return buf[:nv * slot].reshape(nv, slot)[m]
    ok_len = bool(np.array_equal(got_len, want_len))
    h_got = hashlib.sha256(compact(got, got_len).tobytes()).hexdigest()
    h_want = hashlib.sha256(compact(want, want_len).tobytes()).hexdigest()
    return {"blocks": int(nv), "lengths_equal": ok_len, "sha256_equal": h_got == h_want, "sha256": h_got,
            "checker": codec.kind}


class GpuEngine:
    """The product path bench.py measures: device buffers are torch tensors, every codec call goes
    through the C-ABI (csnappy_amd.api).  tests/test_bench_ranks_cpu.py drives run() with an
    engine of the same shape in which the CPU checker stands in for the kernels, so that the
    rank / world plumbing below (block ranges, slowest-rank reduction, the gather) is exercised by
    a 2-rank gloo test before it ever meets eight GPUs."""
    backend = "nccl"

    def __init__(self, local_rank):
        import torch
        from csnappy_amd import api
        api.require_device()
        torch.cuda.set_device(local_rank)
        self.torch, self.api, self.local_rank = torch, api, local_rank
        self.device = torch.device("cuda", local_rank)

    def init_dist(self, dist):
        dist.init_process_group("nccl", device_id=self.device)

    def generate(self, kind, seed, first, nb, block, urls=None):
        torch = self.torch
        if kind >= 0:
            return self.api.generate(kind, seed, first, nb, block)
        rep = torch.from_numpy(np.frombuffer(urls, dtype=np.uint8).copy()).cuda()
        idx = (torch.arange(nb * block, device="cuda", dtype=torch.int64) + first * block) % len(urls)
        return rep[idx]

    def batch(self, lens):
        # HBM is 288 GB: the workspace is sized for parser launches of up to 8 GiB of input (a 1 GiB
        # batch is one launch either way: 4.1 GiB of workspace; an 8 GiB chunk of the large
        # configurations is one launch instead of eight, whose ramps and tails cost 5-11 %: 33 GiB)
        return self.api.Batch(lens, launch_gib=8)

    def zeros(self, n, dtype):
        return self.torch.zeros(n, dtype=dtype, device="cuda")

    def full(self, n, value, dtype):
        return self.torch.full((n,), value, dtype=dtype, device="cuda")

    def compress(self, src, b, cnt, d_out, p, mode):
        self.api.compress_batch(src, b.d_in_off[:cnt], b.d_in_len[:cnt], b.max_in_len, d_out, b.d_out_off[:cnt],
                                b.d_out_len[:cnt], p, mode, b.d_ws)

    def decompress(self, d_out, b, cnt, d_back, cap, status, produced, mode):
        self.api.decompress_batch(d_out, b.d_out_off[:cnt], b.d_out_len[:cnt], d_back, b.d_in_off[:cnt], cap[:cnt],
                                  status[:cnt], produced[:cnt], mode)

    def sync(self):
        self.torch.cuda.synchronize()

    def timing(self, on):
        if on:
            self.api.get_kernel_timing()
        self.api.set_kernel_timing(on)

    def kernel_times(self):
        return self.api.get_kernel_timing()

    def copy_bandwidth(self):
        return measured_copy_bandwidth(self.torch, 1 << 30)

    def verify(self, d_in, b, d_out, chunk, block, p, mode, nv):
        return verify_against_reference(self.torch, self.api, d_in, b, d_out, chunk, block, p, mode, nv)

    def prepare_gather(self, d_out, b, cnt):
        """The local half of the gather (no collective inside): this rank's compacted stream."""
        from csnappy_amd import shard
        return shard.compact(d_out, b.d_out_off[:cnt], b.d_out_len[:cnt])[0]

    def root_buffer(self, prepared, dist, world):
        """Collective (8 B per rank), then rank 0's one allocation: the buffer the streams are
        assembled in -- made before the ranks agree to enter the exchange, so that a root that has
        no room for it costs the record its `gather` field instead of leaving seven ranks in isend."""
        from csnappy_amd import shard
        sizes = shard.exchange_sizes(prepared, dist, world)
        return shard.root_buffer(sizes, prepared.device) if dist.get_rank() == 0 else None

    def time_gather(self, prepared, root_out, d_out, b, cnt, dist, world):
        from csnappy_amd import shard
        return shard.time_gather_compacted(d_out, b, dist, world, cnt=cnt, dense=prepared, root_out=root_out)


# The other BASELINE.json configurations, measured briefly behind the headline one (rank 0 of a 1-GPU
# run) so that the driver's record holds them too: name -> (workload, GiB, what it stands for)
OTHER_CONFIGS = {
    "config3_urls_1gib_p16": ("urls", 1.0, "configs[2]: 1 GiB urls.10K-replicated, round trip"),
    "config4_page_8gib_p13": ("page", 8.0, "configs[3]: zram-style 4 KiB pages, FRAGMENT mode, p=13 (one GPU's 8 GiB of it)"),
    "config5_low_8gib_p16": ("low", 8.0, "configs[4]: low-entropy synthetic (one 8 GiB chunk of it)"),
}


def measure_other_configs(eng, steps, scale=1.0):
    """A short round-trip measurement of each OTHER_CONFIGS entry on this GPU: `steps` timed steps
    behind one checked step (every block round-trips, outside the timed region).  The record's
    `value`, `config` and `roofline` stay the headline configuration's; this adds
    {name: {value, compress_gibs, decompress_gibs, ratio, roofline_frac, ...}}."""
    torch = eng.torch
    res = {}
    for name, (workload, gib, what) in OTHER_CONFIGS.items():
        kind, seed, block, p, mode, desc = WORKLOADS[workload]
        try:
            nb = max(1, int(gib * scale * 2 ** 30) // block)
            urls = open(os.path.join(ROOT, "tests", "golden", "urls.10K"), "rb").read() if kind < 0 else None
            d_in = eng.generate(kind, seed, 0, nb, block, urls)
            b = eng.batch([block] * nb)
            d_out = eng.zeros(b.out_bytes, torch.uint8)
            d_back = eng.zeros(nb * block, torch.uint8)
            cap = eng.full(nb, block, torch.int32)
            status = eng.full(nb, -99, torch.int32)
            produced = eng.zeros(nb, torch.int32)

            def step():
                eng.compress(d_in, b, nb, d_out, p, mode)
                eng.decompress(d_out, b, nb, d_back, cap, status, produced, mode)
            step()
            eng.sync()
            ok = bool((status == 0).all().item()) and bool(torch.equal(d_back[:nb * block], d_in[:nb * block]))
            comp = int(b.d_out_len[:nb].to(torch.int64).sum().item())
            eng.timing(True)
            eng.sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            eng.sync()
            dt = (time.perf_counter() - t0) / steps
            eng.timing(False)
            kt = {k: ms / steps for k, (ms, _) in eng.kernel_times().items()}
            n = nb * block
            c_ms = kt["snappy_parse_fragments"] + kt["snappy_emit_blocks"]
            d_ms = kt["snappy_decompress_blocks"]
            gibs = lambda ms: round(n / (ms / 1e3) / 2 ** 30, 3) if ms > 0 else None
            res[name] = {
                "what": what, "workload": f"{nb * block / 2 ** 30:g} GiB of {desc}, {block}-byte blocks, "
                                          f"{'STREAM' if mode == 0 else 'FRAGMENT'} mode, table power {p}",
                "value": round(n / dt / 2 ** 30, 3), "unit": "GiB/s", "steps": steps,
                "ms_per_step": round(dt * 1e3, 3), "round_trip_ok": ok,
                "compress_gibs": gibs(c_ms), "decompress_gibs": gibs(d_ms),
                "ratio": round(comp / n, 6),
                "kernel_ms": {k: round(v, 4) for k, v in kt.items()},
                # the compress operation's algorithmic bytes (N_in + C_out) over its kernels' time, of the 8 TB/s roof
                "roofline_frac": round((n + comp) / (c_ms / 1e3) / 1e9 / HBM_PEAK_GBS, 6) if c_ms > 0 else None,
                "roofline_frac_decompress": round((n + comp) / (d_ms / 1e3) / 1e9 / HBM_PEAK_GBS, 6) if d_ms > 0 else None,
            }
            del d_in, b, d_out, d_back, cap, status, produced
        except Exception as e:  # noqa: BLE001 -- a secondary figure must never take the headline line down
            res[name] = {"what": what, "error": f"{type(e).__name__}: {e}"[:200]}
        if hasattr(torch, "cuda") and torch.cuda.is_available():
            torch.cuda.empty_cache()
    return res


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="text")
    ap.add_argument("--gib", type=float, default=1.0, help="uncompressed GiB per GPU")
    ap.add_argument("--p", type=int, default=None, help="table power (default per workload)")
    ap.add_argument("--block", type=int, default=None, help="block bytes (default per workload)")
    ap.add_argument("--chunk-gib", type=float, default=8.0,
                    help="blocks are processed in chunks of this many GiB (bounds the output/workspace "
                         "memory of very large batches; the whole input stays resident)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify-gib", type=float, default=1.0,
                    help="outside the timed region, compare the compressed bytes and lengths of the first "
                         "this-many GiB of blocks with the CPU reference (0 = round-trip check only)")
    ap.add_argument("--gather", dest="gather", action="store_true", default=None,
                    help="also time the RCCL gather of the compacted per-rank streams to rank 0 (reported "
                         "separately; never part of `value`).  Default: on when there is more than one rank")
    ap.add_argument("--no-gather", dest="gather", action="store_false")
    ap.add_argument("--other-configs", dest="other_configs", action="store_true", default=None,
                    help="behind the headline measurement, measure the other BASELINE.json configurations briefly "
                         "(10 steps each) and add them to the record as `other_configs`.  Default: on for the "
                         "default 1-GPU text run, off otherwise")
    ap.add_argument("--no-other-configs", dest="other_configs", action="store_false")
    ap.add_argument("--other-scale", type=float, default=1.0, help=argparse.SUPPRESS)  # tests: shrink the other configs
    ap.add_argument("--engine", default=None, help=argparse.SUPPRESS)  # tests: module:Class standing in for GpuEngine
    return ap.parse_args(argv)


def run(args, eng, dist, rank, world):
    """One bench run on this rank.  -> the JSON record on rank 0, None elsewhere."""
    from csnappy_amd import shard
    torch = eng.torch
    kind, seed, block, p, mode, desc = WORKLOADS[args.workload]
    block = args.block or block
    p = args.p or p
    nb = max(1, int(args.gib * 2 ** 30) // block)  # blocks per rank
    first, _ = shard.block_range(nb * world, rank, world)

    # ---- input resident in HBM -------------------------------------------------------------------
    urls = None
    if kind < 0:
        urls = open(os.path.join(ROOT, "tests", "golden", "urls.10K"), "rb").read()
    d_in = eng.generate(kind, seed, first, nb, block, urls)
    # the batch is processed in chunks of <= chunk_gib (one chunk for the default 1 GiB workload)
    cb = max(1, min(nb, int(args.chunk_gib * 2 ** 30) // block))  # blocks per chunk
    chunks = [(lo, min(cb, nb - lo)) for lo in range(0, nb, cb)]
    b = eng.batch([block] * cb)
    d_out = eng.zeros(b.out_bytes, torch.uint8)
    d_back = eng.zeros(cb * block, torch.uint8)
    cap = eng.full(cb, block, torch.int32)
    status = eng.full(cb, -99, torch.int32)
    produced = eng.zeros(cb, torch.int32)
    comp_total = eng.zeros(1, torch.int64)
    eng.sync()

    def run_chunk(lo, cnt, check=False):
        src = d_in[lo * block:(lo + cnt) * block]
        eng.compress(src, b, cnt, d_out, p, mode)
        eng.decompress(d_out, b, cnt, d_back, cap, status, produced, mode)
        if check:  # outside the timed region only
            eng.sync()
            assert (status[:cnt] == 0).all().item(), "decompress reported an error"
            assert torch.equal(d_back[:cnt * block], src), "round trip differs from the input"
            comp_total.add_(b.d_out_len[:cnt].to(torch.int64).sum())

    def step(check=False):
        for lo, cnt in chunks:
            run_chunk(lo, cnt, check)

    def barrier():
        eng.sync()
        if dist is not None:
            dist.barrier()
        eng.sync()

    # results are checked outside the timed region: every block of every chunk round-trips
    step(check=True)
    comp_bytes = int(comp_total.item())
    # ... and (rank 0) the first --verify-gib of blocks are bit-exact against the CPU reference:
    # every compressed length, and the sha256 of the compacted stream
    bit_exact = None
    if rank == 0 and args.verify_gib > 0:
        bit_exact = eng.verify(d_in, b, d_out, chunks[0], block, p, mode,
                               min(chunks[0][1], max(1, int(args.verify_gib * 2 ** 30) // block)))
    for _ in range(max(0, args.warmup - 1)):
        step()
    eng.sync()

    # ---- timed region: exactly K steps ----------------------------------------------------------
    eng.timing(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    eng.timing(False)
    kt = eng.kernel_times()

    t_max, t_min, ranks_seen = elapsed, elapsed, 1
    n_bytes = nb * block
    kt_ms = [kt[k][0] for k in sorted(kt)]
    if dist is not None:
        ranks_seen = dist.get_world_size()
        t = torch.tensor([elapsed, -elapsed] + kt_ms, dtype=torch.float64, device=eng.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)  # slowest rank, also per kernel
        t_max, t_min = t[0].item(), -t[1].item()
        kt = {k: (t[2 + i].item(), kt[k][1]) for i, k in enumerate(sorted(kt))}
    # a "launch" below is one batch call of the C-ABI (per chunk); ms_per_step sums the chunks;
    # with several ranks every figure is that of the SLOWEST rank
    kernels = {k: {"avg_ms": round(ms / max(c, 1), 4), "launches": c,
                   "ms_per_step": round(ms / args.steps, 4)} for k, (ms, c) in kt.items()}

    # the one collective of the path: the final stream to rank 0 (throughput is reported
    # both without and with it).  On by default as soon as there is more than one rank.
    gather = None
    gather_error = None
    want_gather = args.gather if args.gather is not None else world > 1
    if want_gather and dist is not None:
        # (outside the timed region; a gather that fails must not cost the run its throughput line.
        # The part that can fail on ONE rank -- allocations, the compaction -- runs first and on its own;
        # the ranks then agree (all_reduce MIN of an ok flag) and enter the exchange together or not at
        # all: a rank that raised before the collective would otherwise leave the others waiting in it.)
        def agreed(err):
            """True when no rank has an error; a collective that fails is an error too."""
            ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=eng.device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            return bool(ok.item())

        prepared, root_out, err = None, None, None
        try:
            try:
                prepared = eng.prepare_gather(d_out, b, chunks[-1][1])  # the last chunk's output
            except Exception as e:  # noqa: BLE001 -- reported in the record, the measurement stands
                err = f"{type(e).__name__}: {e}"[:200]
            if agreed(err):
                # rank 0's assembly buffer: the one allocation only one rank makes, so it is made
                # (and agreed on) before anybody posts a send
                try:
                    root_out = eng.root_buffer(prepared, dist, world)
                except Exception as e:  # noqa: BLE001
                    err = f"{type(e).__name__}: {e}"[:200]
                if agreed(err):
                    gather = eng.time_gather(prepared, root_out, d_out, b, chunks[-1][1], dist, world)
            if gather is None:
                gather_error = err or "another rank could not prepare its stream"
        except Exception as e:  # noqa: BLE001 -- a collective itself failed (RCCL error): the line is still printed
            gather_error = f"{type(e).__name__}: {e}"[:200]

    if rank != 0:
        return None

    # ---- roofline of the dominant operation (this GPU) -------------------------------------------
    # algorithmic bytes per batch: compress N_in + C_out, decompress C_in + N_out.
    # Compress is two kernels (the parser, which bounds it, and the emit kernel that writes C); the
    # figure is over their summed duration, i.e. the whole compress operation.
    ops = {"compress": (("snappy_parse_fragments", "snappy_emit_blocks"), n_bytes + comp_bytes),
           "decompress": (("snappy_decompress_blocks",), comp_bytes + n_bytes)}
    op_ms = {o: sum(kernels[k]["ms_per_step"] for k in ks) for o, (ks, _) in ops.items()}
    dom = max(ops, key=lambda o: op_ms[o])
    dom_kernels, dom_alg = ops[dom]
    nch = len(chunks)
    dom_s = op_ms[dom] / 1e3
    achieved = dom_alg / dom_s / 1e9 if dom_s > 0 else 0.0
    traffic = load_measured_traffic(args.workload, p, dom_kernels) if (nch == 1 and args.gib == 1.0
                                                                       and args.block is None) else None
    peak_measured = eng.copy_bandwidth()
    roofline = {"bound": "hbm", "kernel": dom_kernels[0], "operation": dom, "kernels_of_operation": list(dom_kernels),
                "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                "peak_measured": peak_measured,
                "frac_of_measured": round(achieved / peak_measured, 6) if peak_measured else None,
                "algorithmic_bytes_per_launch": dom_alg // nch,
                "avg_launch_ms": {k: kernels[k]["avg_ms"] for k in dom_kernels},
                "operation_ms_per_step": round(op_ms[dom], 4),
                "launches_per_step": nch, "traffic": traffic,
                # (not measured by this run: PMC counters need rocprofv3 passes of their own)
                "traffic_source": "profiles/pmc_traffic.json (rocprofv3 PMC, separate run)" if traffic is not None
                else None}

    gibs = lambda ms: round(n_bytes * world / (ms / 1e3) / 2 ** 30, 3) if ms > 0 else None
    out = {
        "metric": "GiB/s compress+decompress on 64KiB blocks" if block == 65536 else
                  f"GiB/s compress+decompress on {block}-byte blocks",
        "value": round(n_bytes * world * args.steps / t_max / 2 ** 30, 4),
        "unit": "GiB/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(t_max / args.steps * 1e3, 4),
        "n_ranks_seen": ranks_seen,
        "rank_ms_per_step": {"min": round(t_min / args.steps * 1e3, 4), "max": round(t_max / args.steps * 1e3, 4)},
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic" if kind >= 0 else "urls.10K replicated",
        "config": {"workload": f"{args.gib:g} GiB per GPU of {desc}, {block}-byte blocks, "
                               f"{'STREAM' if mode == 0 else 'FRAGMENT'} mode, table power {p}, "
                               "compress then decompress (round trip), inputs resident in HBM",
                   "block_bytes": block, "blocks_per_gpu": nb, "table_power": p,
                   "mode": "STREAM" if mode == 0 else "FRAGMENT", "seed": hex(seed),
                   "chunks_per_step": len(chunks),
                   "sharding": f"block ranges, {world} rank(s), no data-path collective"},
        "compressed_ratio": round(comp_bytes / n_bytes, 6),
        "bit_exact_blocks": (bit_exact["blocks"] if bit_exact and bit_exact["lengths_equal"]
                             and bit_exact["sha256_equal"] else 0) if bit_exact else None,
        "bit_exact": bit_exact,
        "compress_gibs": gibs(kernels["snappy_parse_fragments"]["ms_per_step"]
                              + kernels["snappy_emit_blocks"]["ms_per_step"]),
        "decompress_gibs": gibs(kernels["snappy_decompress_blocks"]["ms_per_step"]),
        "kernels": kernels,
        "roofline": roofline,
    }
    if gather_error is not None:
        out["gather_error"] = gather_error
    if gather is not None:
        # the same round trip with the gather of the final stream added to every step's time
        # (the last chunk's stream stands for the step's: one chunk in the default workload)
        out["gather"] = gather
        # (scaled by blocks, not by chunks: the last chunk may be a short one)
        out["value_with_gather"] = round(n_bytes * world / (t_max / args.steps
                                                            + gather["ms"] / 1e3 * nb / chunks[-1][1]) / 2 ** 30, 4)
    want_other = args.other_configs if args.other_configs is not None else (
        world == 1 and args.workload == "text" and args.gib == 1.0 and args.block is None and args.p is None)
    if want_other and world == 1:
        # the headline buffers are not needed any more: give their HBM back first
        del d_in, d_out, d_back, cap, status, produced, b
        if hasattr(torch, "cuda") and torch.cuda.is_available():
            torch.cuda.empty_cache()
        out["other_configs"] = measure_other_configs(eng, 10, args.other_scale)
    if world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(kind, seed, block, p, mode, nb, args.cpu_seconds, urls)
        except Exception as e:  # the baseline must never take the bench line down
            out["cpu_baseline"] = {"value": None, "unit": "GiB/s", "cores": os.cpu_count(),
                                   "kind": "port", "sample": f"failed: {e!r}"}
    return out


def launch_ranks(argv, gpus):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD process
    (torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1), relay its output and
    return its exit code.  Called before anything in this process has touched the GPU, and never
    an exec: a process that has initialised HIP must not be replaced."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1",
           "--nnodes=1", f"--nproc-per-node={gpus}", os.path.abspath(__file__)] + list(argv)
    log("bench.py: starting %d ranks: %s" % (gpus, " ".join(cmd)))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # RCCL across processes needs dmabuf IPC here
    return subprocess.run(cmd, env=env).returncode


def load_engine(spec, local_rank):
    """GpuEngine, or (tests only: --engine module:Class) a stand-in of the same shape."""
    if not spec:
        return GpuEngine(local_rank)
    import importlib
    mod, _, cls = spec.partition(":")
    return getattr(importlib.import_module(mod), cls)()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(argv, args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE")
    eng = load_engine(args.engine, local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run (also with 1 rank)
        import torch.distributed as dist
        eng.init_dist(dist)
    out = run(args, eng, dist, rank, world)
    if out is not None:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
