#!/usr/bin/env python3
"""Digest rocprofv3 outputs (tools/profile_gpu.sh) into one small JSON for profiles/.

Kernel durations come from the --kernel-trace --stats pass.  HBM traffic comes from the two PMC
passes and is corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes:
FETCH_SIZE / WRITE_SIZE are in KiB, and on gfx950 FETCH_SIZE reports half the bytes of a wide
(16 B/lane) coalesced read -- which is how the compress kernel stages its window -- so it is
doubled; WRITE_SIZE is taken as is (uncalibrated, per the guide)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out, args = sys.argv[1], sys.argv[2]


def rows(pattern):
    for path in glob.glob(os.path.join(out, pattern), recursive=True):
        with open(path, newline="") as f:
            yield from csv.DictReader(f)


summary = {"bench_args": args, "kernels": {}, "pmc": {}, "per_batch": {}}


def family(name):
    """compress runs as one or more chunk launches of one of its instantiations per batch call"""
    if name.startswith("snappy_parse_fragments"):
        return "snappy_parse_fragments"
    return "snappy_emit_blocks" if name.startswith("snappy_emit_") else name  # (pages: snappy_emit_pages)
for r in rows("stats/**/*kernel_stats.csv"):
    name = r.get("Name", "")
    if "snappy" in name or "workload" in name:
        summary["kernels"][name.split("(")[0]] = {
            "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "total_ns": float(r["TotalDurationNs"]),
            "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"]), "percent": float(r["Percentage"])}

for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = defaultdict(lambda: [0.0, 0])
    for r in rows(f"{counter[:5].lower()}/**/*counter_collection.csv"):
        if r.get("Counter_Name") != counter:
            continue
        k = r.get("Kernel_Name", "").split("(")[0]
        if "snappy" not in k:
            continue
        acc[k][0] += float(r["Counter_Value"])
        acc[k][1] += 1
    for k, (tot, n) in acc.items():
        summary["pmc"].setdefault(k, {})[counter + "_KiB_per_launch"] = tot / max(n, 1)
        summary["pmc"][k]["launches_" + counter] = n

for k, v in summary["pmc"].items():
    if "FETCH_SIZE_KiB_per_launch" in v and "WRITE_SIZE_KiB_per_launch" in v:
        v["hbm_bytes_per_launch_corrected"] = int((2 * v["FETCH_SIZE_KiB_per_launch"]
                                                   + v["WRITE_SIZE_KiB_per_launch"]) * 1024)
try:
    summary["bench_line"] = json.loads(open(os.path.join(out, "bench_stats.json")).read().strip().splitlines()[-1])
except Exception as e:  # noqa
    summary["bench_line"] = f"unreadable: {e!r}"

# per BATCH CALL figures (what bench.py's HIP events measure).  A compress batch call launches the
# parser up to three times (two dense table sizes, then the global table; two of them under one
# kernel name) and the emit kernel once, per chunk; bench.py calls the compress batch once more
# than the decompress batch (the bit-exactness check).  So: batch calls = decompress launches
# (+ 1 for compress), and every kernel's TOTAL over the run is divided by that.
try:
    def batches_of(fam, counts):
        # decompress is one launch per batch call; compress launches once per chunk of 32 768
        # fragments (several chunks for a GiB of 4 KiB pages) and is called once more than
        # decompress (the bit-exactness check)
        dec = max(counts.get("snappy_decompress_blocks", 0), 1)
        return dec + 1 if fam in ("snappy_parse_fragments", "snappy_emit_blocks") else max(counts.get(fam, 0), 1)
    calls = {n: k["calls"] for n, k in summary["kernels"].items()}
    for name, k in summary["kernels"].items():
        f = summary["per_batch"].setdefault(family(name), {"ms": 0.0, "launches_per_batch": 0.0})
        b = batches_of(family(name), calls)
        f["ms"] += k["total_ns"] / b / 1e6
        f["launches_per_batch"] += k["calls"] / b
    for counter, key in (("FETCH_SIZE", "fetch_bytes_raw"), ("WRITE_SIZE", "write_bytes_raw")):
        counts = {n: v.get("launches_" + counter, 0) for n, v in summary["pmc"].items()}
        for name, v in summary["pmc"].items():
            if counter + "_KiB_per_launch" not in v:
                continue
            f = summary["per_batch"].setdefault(family(name), {})
            total = v[counter + "_KiB_per_launch"] * v["launches_" + counter] * 1024
            f[key] = f.get(key, 0) + int(total / batches_of(family(name), counts))
    for f in summary["per_batch"].values():
        if "fetch_bytes_raw" in f and "write_bytes_raw" in f:
            f["hbm_bytes_corrected"] = 2 * f["fetch_bytes_raw"] + f["write_bytes_raw"]
except Exception as e:  # noqa
    summary["per_batch"] = f"unavailable: {e!r}"
print(json.dumps(summary, indent=1))
