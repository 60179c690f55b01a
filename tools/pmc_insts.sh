#!/bin/bash
# Run ON THE GPU BOX (via gpurun): instruction-mix counters per kernel, one PMC pass per group.
#   bash tools/pmc_insts.sh <tag> [bench args]   ->  gpurun_out/insts_<tag>/summary.txt
set -u
tag=$1; shift
out=gpurun_out/insts_$tag
mkdir -p $out
export TMPDIR=/tmp
args="--steps 2 --warmup 1 --no-cpu-baseline --no-other-configs $*"
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $out/g$i -o run -- python3 bench.py $args > $out/bench_$i.json 2> $out/bench_$i.err
done
python3 - "$out" > $out/summary.txt <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(int))
for path in glob.glob(os.path.join(out, "g*/**/*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path, newline="")):
        k = r["Kernel_Name"].split("(")[0]
        if "snappy" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        print(f"  {c:24s} {acc[k][c]/n[k][c]:16.0f} per launch  ({n[k][c]} launches)")
PY
cat $out/summary.txt
