#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:  bash tools/profile_gpu.sh <tag> [bench args]
# Produces under gpurun_out/prof_<tag>/:
#   stats/   rocprofv3 --kernel-trace --stats of `python3 bench.py <args>`
#   fetch/   rocprofv3 --pmc FETCH_SIZE   (own pass: FETCH_SIZE and WRITE_SIZE do not fit one pass)
#   write/   rocprofv3 --pmc WRITE_SIZE
# and a digest gpurun_out/prof_<tag>/summary.json (tools/profile_digest.py) that is what gets
# copied into profiles/.
set -u
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
args="--steps 3 --warmup 1 --no-cpu-baseline --no-other-configs $*"
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o run -- python3 bench.py $args > $out/bench_stats.json 2> $out/bench_stats.err
timeout 420 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o run -- python3 bench.py $args > $out/bench_fetch.json 2> $out/bench_fetch.err
timeout 420 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o run -- python3 bench.py $args > $out/bench_write.json 2> $out/bench_write.err
python3 tools/profile_digest.py $out "$args" > $out/summary.json 2> $out/digest.err
find $out -name '*.csv' -size +8M -delete
ls -R $out | head -40
cat $out/summary.json
