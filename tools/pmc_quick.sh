#!/bin/bash
# Run ON THE GPU BOX: one PMC pass over a short bench run, per-kernel averages.
#   bash tools/pmc_quick.sh <tag> "<counters>" [bench args]  ->  gpurun_out/pmcq_<tag>.txt
set -u
tag=$1; ctrs=$2; shift 2
out=gpurun_out/pmcq_$tag
mkdir -p $out
export TMPDIR=/tmp
timeout ${PMCQ_TIMEOUT:-150} rocprofv3 --pmc $ctrs --output-format csv -d $out/g -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --verify-gib 0 "$@" > $out/bench.json 2> $out/bench.err
python3 - "$out" > gpurun_out/pmcq_$tag.txt <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(int))
for path in glob.glob(os.path.join(out, "g/**/*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path, newline="")):
        k = r["Kernel_Name"].split("(")[0]
        if "snappy" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        print(f"  {c:24s} {acc[k][c]/n[k][c]:16.0f} per launch  ({n[k][c]} launches)")
PY
rm -rf $out/g
cat gpurun_out/pmcq_$tag.txt
