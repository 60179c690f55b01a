for w in 1 4 16; do
  echo -n "wgs $w distinct: "; CSNAPPY_HIP_WGS_PER_CU=$w timeout 120 python tools/time_same.py 0 2>&1 | tail -1
  echo -n "wgs $w same:     "; CSNAPPY_HIP_WGS_PER_CU=$w timeout 120 python tools/time_same.py 1 2>&1 | tail -1
done
