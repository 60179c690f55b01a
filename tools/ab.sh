#!/bin/bash
# Run ON THE GPU BOX: time_parse.py for the default library and for variant builds, interleaved, twice.
#   bash tools/ab.sh "<workloads>" <variant> [<variant> ...]
ws=$1; shift
for rep in 1 2; do
  for w in $ws; do
    python tools/time_parse.py $w 1024 2>&1 | grep -v amdgpu
    for v in "$@"; do CSNAPPY_AMD_LIB=$PWD/build/var/$v/libcsnappy.so python tools/time_parse.py $w 1024 2>&1 | grep -v amdgpu; done
  done
done
