for cap in 3072 4096 5120 6144; do for sp in 2048 4096; do echo -n "cap $cap spill $sp: "; CSNAPPY_HIP_DENSE_CAP=$cap CSNAPPY_HIP_SPILL_CAP=$sp python tools/time_emit.py 2>&1 | tail -1; done; done
