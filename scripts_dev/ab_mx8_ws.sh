#!/bin/bash
# GPU box: MX-fp8 GEMM in loader / consumer form (config 2) -- tests, alone timings, cfg 5 bf16 vs fp8 step
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_mx8.py -x -q -m gpu > gpurun_out/mx_ws_tests.log 2>&1; rc=$?; tail -n 3 gpurun_out/mx_ws_tests.log; [ $rc = 0 ] || exit 2
timeout -k 10 300 python scripts_dev/bench_mx8.py > gpurun_out/mx_ws_alone.log 2>&1 || { tail -5 gpurun_out/mx_ws_alone.log; exit 3; }
grep "^M=" gpurun_out/mx_ws_alone.log | cut -c1-260
for i in 1 2; do
  for w in cfg5 cfg5_fp8; do
    timeout -k 10 200 python bench.py --workload $w --steps 30 --no-cpu-baseline > gpurun_out/mx_ws_$w$i.json 2> gpurun_out/mx_ws_$w$i.err || exit 4
    python3 -c "
import json
d=json.loads(open('gpurun_out/mx_ws_$w$i.json').read().strip().splitlines()[-1]); print('$w run $i', d['ms_per_step'], d['value'], d['roofline']['device_ms_per_step_by_class'])"
  done
done
