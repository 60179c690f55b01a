"""bench.py's rank / world path on CPU: two gloo ranks run bench.run() with an engine in which the
CPU checker stands in for the HIP kernels (bench.GpuEngine is the product path; this file is a
test of the plumbing around it -- block ranges per rank, the slowest-rank reduction of the wall
and per-kernel times, n_ranks_seen, the default-on gather of the compacted streams to rank 0 and
the record's shape -- so that a first 8-GPU driver run does not die in it)."""
import hashlib
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import bench  # noqa: E402
import oracle  # noqa: E402
from csnappy_amd import api  # noqa: E402

from tests.cpu_engine import OracleEngine  # noqa: E402

BLOCK, NB = 65536, 24  # blocks per rank


def _args(world):
    return bench.parse_args(["--gpus", str(world), "--steps", "2", "--warmup", "1", "--gib", str(NB * BLOCK / 2 ** 30),
                             "--no-cpu-baseline", "--verify-gib", "0"])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eng = OracleEngine()
        rec = bench.run(_args(world), eng, dist, rank, world)
        if rank == 0:
            ret["rec"] = rec
            ret["gathered_sha"] = hashlib.sha256(eng.rooted.numpy().tobytes()).hexdigest()
        else:
            assert rec is None
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_through_bench_run():
    world = 2
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        rec, gathered = ret["rec"], ret["gathered_sha"]
    assert rec["n_gpus"] == 2 and rec["n_ranks_seen"] == 2 and rec["scaling"] == "weak"
    assert rec["config"]["blocks_per_gpu"] == NB and rec["value"] > 0
    assert rec["rank_ms_per_step"]["min"] <= rec["rank_ms_per_step"]["max"] == rec["ms_per_step"]
    assert set(rec["kernels"]) == {"snappy_parse_fragments", "snappy_emit_blocks", "snappy_decompress_blocks"}
    assert rec["kernels"]["snappy_decompress_blocks"]["launches"] == 2  # one chunk per step, two steps
    assert rec["roofline"]["operation"] in ("compress", "decompress") and rec["roofline"]["frac"] > 0
    # more than one rank: the gather of the final stream is timed by default, next to `value`
    assert rec["gather"]["gathered_bytes"] > 0 and 0 < rec["value_with_gather"] < rec["value"]
    assert "cpu_baseline" not in rec  # rank 0 at N = 1 only
    # the stream rank 0 assembled = the blocks of both ranks' ranges, in block order
    kind, seed, block, p, mode, _ = bench.WORKLOADS["text"]
    host = api.generate_host(kind, seed, 0, world * NB, block)
    want = b"".join(oracle.Port().compress_blocks(host, block, p, mode))
    assert rec["gather"]["gathered_bytes"] == len(want) and gathered == hashlib.sha256(want).hexdigest()


def test_one_rank_without_a_process_group():
    rec = bench.run(_args(1), OracleEngine(), None, 0, 1)
    assert rec["n_gpus"] == 1 and rec["n_ranks_seen"] == 1 and "gather" not in rec and rec["vs_baseline"] is None
    assert rec["metric"] == "GiB/s compress+decompress on 64KiB blocks" and rec["unit"] == "GiB/s"
    assert 0.3 < rec["compressed_ratio"] < 0.7


def _bench_cli(extra_env=None):
    """`python bench.py --gpus 2 ...` exactly as the driver types it for N = 1 -- no launcher around it."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = os.path.dirname(HERE) + os.pathsep + env.get("PYTHONPATH", "")
    env.update(extra_env or {})
    cmd = [sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--gib", str(NB * BLOCK / 2 ** 30), "--no-cpu-baseline", "--verify-gib", "0",
           "--engine", "tests.cpu_engine:OracleEngine"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # ONE JSON line, from rank 0
    return json.loads(lines[0]), r.stderr


@pytest.mark.timeout(300)
def test_bench_py_starts_its_own_ranks():
    """`bench.py --gpus 2` with no RANK in the environment starts two ranks itself (a child
    torch.distributed.run; the parent never touches a GPU) and relays rank 0's record."""
    rec, err = _bench_cli()
    assert "torch.distributed.run" in err and "--nproc-per-node=2" in err
    assert rec["n_gpus"] == 2 and rec["n_ranks_seen"] == 2 and rec["value"] > 0
    assert rec["gather"]["gathered_bytes"] > 0 and "gather_error" not in rec


@pytest.mark.timeout(300)
def test_a_rank_that_cannot_prepare_its_stream_costs_the_gather_not_the_record():
    """One rank fails in the local half of the gather: the ranks agree not to enter the exchange
    (nobody is left waiting in a collective) and rank 0 still prints the throughput record."""
    rec, _ = _bench_cli({"CSNAPPY_TEST_FAIL_PREPARE_ON_RANK": "1"})
    assert rec["n_ranks_seen"] == 2 and rec["value"] > 0
    assert "gather" not in rec and "another rank" in rec["gather_error"]
    rec, _ = _bench_cli({"CSNAPPY_TEST_FAIL_PREPARE_ON_RANK": "0"})
    assert "gather" not in rec and "MemoryError" in rec["gather_error"]
    # rank 0 has no room for the assembly buffer (the one allocation only one rank makes): it is made
    # before the ranks agree to enter the exchange, so nobody is left in isend
    rec, _ = _bench_cli({"CSNAPPY_TEST_FAIL_ROOT_BUFFER": "1"})
    assert rec["n_ranks_seen"] == 2 and rec["value"] > 0
    assert "gather" not in rec and "root buffer" in rec["gather_error"]


@pytest.mark.timeout(300)
def test_the_record_holds_the_other_baseline_configurations():
    """`bench.py --gpus 1` measures BASELINE.json's other configurations briefly behind the headline
    one; here with the CPU stand-in engine and the sizes shrunk, for the record's shape."""
    args = bench.parse_args(["--gpus", "1", "--steps", "1", "--warmup", "1", "--gib", str(8 * BLOCK / 2 ** 30),
                             "--no-cpu-baseline", "--verify-gib", "0", "--other-configs", "--other-scale",
                             str(4 * BLOCK / 2 ** 30)])
    rec = bench.run(args, OracleEngine(), None, 0, 1)
    assert rec["value"] > 0 and "text" in rec["config"]["workload"].lower()
    oc = rec["other_configs"]
    assert set(oc) == {"config3_urls_1gib_p16", "config4_page_8gib_p13", "config5_low_8gib_p16"}
    for name, r in oc.items():
        assert "error" not in r, (name, r)
        assert r["round_trip_ok"] is True and r["value"] > 0 and 0 < r["ratio"] < 1.2
        for key in ("compress_gibs", "decompress_gibs", "roofline_frac", "kernel_ms", "workload", "steps"):
            assert key in r, (name, key)
    assert "4096-byte" in oc["config4_page_8gib_p13"]["workload"] and "FRAGMENT" in oc["config4_page_8gib_p13"]["workload"]
