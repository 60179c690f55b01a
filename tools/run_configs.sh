#!/bin/bash
# Run ON THE GPU BOX: the BASELINE.json configs that fit one GPU, one bench line each.
#   bash tools/run_configs.sh <outdir>
set -u
out=${1:-gpurun_out/results}
mkdir -p $out
python3 bench.py                                            > $out/config2_text_1gib_p16.json   2> $out/config2.err
python3 bench.py --no-cpu-baseline --p 15                   > $out/config2_text_1gib_p15.json   2>> $out/config2.err
python3 bench.py --no-cpu-baseline --workload urls          > $out/config3_urls_1gib_p16.json   2> $out/config3.err
python3 bench.py --no-cpu-baseline --workload page --gib 16 > $out/config4_page_16gib_p13.json  2> $out/config4.err
python3 bench.py --no-cpu-baseline --workload low --gib 64 --steps 3 --warmup 1 > $out/config5_low_64gib_p16.json 2> $out/config5.err
for f in $out/*.json; do python3 - "$f" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1].split("/")[-1], "value", d["value"], "compress", d["compress_gibs"], "decompress", d["decompress_gibs"],
      "ratio", d["compressed_ratio"], "roofline", d["roofline"]["frac"])
PY
done
