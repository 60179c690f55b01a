#!/bin/bash
# Run ON THE GPU BOX: per-kernel average durations of one short bench run (rocprofv3 --kernel-trace --stats).
#   bash tools/kstats.sh <tag> [bench args]  ->  gpurun_out/kstats_<tag>.txt
set -u
tag=$1; shift
out=gpurun_out/kstats_$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/g -o run -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --verify-gib 0 "$@" > $out/bench.json 2> $out/bench.err
f=$(find $out/g -name '*kernel_stats.csv' | head -1)
python3 - "$f" > gpurun_out/kstats_$tag.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1], newline="")))
for r in rows:
    if "snappy" in r["Name"]:
        print(f"{r['Name'].split('(')[0]:44s} calls {r['Calls']:>4s}  avg {float(r['AverageNs'])/1e6:9.3f} ms  total {float(r['TotalDurationNs'])/1e6:9.3f} ms")
PY
cp "$f" gpurun_out/kstats_${tag}_kernel_stats.csv 2>/dev/null
rm -rf $out/g
cat gpurun_out/kstats_$tag.txt
