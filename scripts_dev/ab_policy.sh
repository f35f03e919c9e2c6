#!/bin/bash
# GPU box: does the tile-cost model of pick_config help?  cfg 2 / cfg 4 / cfg 5 steps, two interleaved rounds each
set -o pipefail
mkdir -p gpurun_out
for i in 1 2; do
  for w in cfg2 cfg4 cfg5 cfg5_fp8; do
    timeout -k 10 200 python bench.py --workload $w --steps 30 --no-cpu-baseline --no-roofline 2> gpurun_out/pol_$w$i.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w run $i', d['ms_per_step'])" || exit 4
  done
done
