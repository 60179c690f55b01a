set -u
out=gpurun_out/kstats_stream
mkdir -p $out
export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/g -o run -- python3 tools/one_stream.py > $out/run.log 2>&1
f=$(find $out/g -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1], newline="")):
    if "snappy" in r["Name"]:
        print(f"{r['Name'].split('(')[0]:40s} calls {r['Calls']:>4s}  avg {float(r['AverageNs'])/1e6:9.3f} ms")
PY
rm -rf $out/g
