for cap in 5120 4992 4864 4800 4736 4672; do echo -n "cap $cap: "; CSNAPPY_HIP_DENSE_CAP=$cap python tools/time_emit.py 2>&1 | tail -1; done
