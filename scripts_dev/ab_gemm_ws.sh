#!/bin/bash
# GPU box: config 9 (loader / consumer waves) -- sweep against configs 0, 1, 2, GEMM tests under force, step A/B of HRIEMO_GEMM_WS
set -o pipefail
mkdir -p gpurun_out
CFGS=0,1,2,9 timeout -k 10 300 python scripts_dev/bench_gemm.py > gpurun_out/ws_sweep.log 2>&1 || { tail -5 gpurun_out/ws_sweep.log; exit 2; }
tail -n 44 gpurun_out/ws_sweep.log
HRIEMO_GEMM_WS=1 timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -x -q -m gpu -k "gemm or train_step or determin or dropout_equals or golden or shared_input" > gpurun_out/ws_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/ws_tests.log; tail -n 3 gpurun_out/ws_tests.log
for i in 1 2 3; do
  for v in 0 1; do
    HRIEMO_GEMM_WS=$v timeout -k 10 200 python bench.py --steps 40 --no-cpu-baseline --no-roofline 2> gpurun_out/ws_b$v$i.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('GEMM_WS=$v run $i:', d['ms_per_step'], d['ms_per_step_events']['median'])" || exit 4
  done
done
