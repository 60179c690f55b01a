for v in base cur base cur; do
  if [ $v = base ]; then export CSNAPPY_AMD_LIB=build/var/base/libcsnappy.so; else unset CSNAPPY_AMD_LIB; fi
  python3 bench.py --no-cpu-baseline --no-other-configs --workload low --gib 8 --steps 3 --warmup 1 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['value'], d['compress_gibs'], {k:round(v['ms_per_step']/8,3) for k,v in d['kernels'].items()})"
done
