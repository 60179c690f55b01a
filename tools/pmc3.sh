#!/bin/bash
# Run ON THE GPU BOX: instruction counters of the parser for the default library and variant builds.
#   bash tools/pmc3.sh <variant> [<variant> ...]
for v in default "$@"; do
  if [ $v = default ]; then unset CSNAPPY_AMD_LIB; else export CSNAPPY_AMD_LIB=$PWD/build/var/$v/libcsnappy.so; fi
  bash tools/pmc_quick.sh i_$v "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" > /dev/null 2>&1
  echo "== $v"; grep -A7 "parse_fragments" gpurun_out/pmcq_i_$v.txt
done
