#!/bin/bash
# Run ON THE GPU BOX: parser time against the fragments in flight per CU (CSNAPPY_HIP_WGS_PER_CU), 256 MiB.
#   bash tools/sweep_wgs.sh "<workloads>" [lib ...]
ws=$1; shift
for w in $ws; do
  for lib in default "$@"; do
    for k in 1 2 4 8 12 16; do
      if [ $lib = default ]; then L=""; else L="CSNAPPY_AMD_LIB=$PWD/build/var/$lib/libcsnappy.so"; fi
      echo -n "wgs_per_cu=$k "
      env $L CSNAPPY_HIP_WGS_PER_CU=$k python tools/time_parse.py $w 256 2>&1 | grep -v amdgpu
    done
  done
done
