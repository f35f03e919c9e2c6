#!/bin/bash
# GPU box: kernel test + alone timings + same-box step A/B of HRIEMO_FUSE_LN (Linear + LayerNorm in one kernel)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k gemm_ln > gpurun_out/fl_ktest.log 2>&1; rc=$?; tail -n 5 gpurun_out/fl_ktest.log; [ $rc = 0 ] || exit 2
timeout -k 10 300 python scripts_dev/bench_gemm_ln.py > gpurun_out/fl_alone.log 2>&1 || { tail -5 gpurun_out/fl_alone.log; exit 3; }
cat gpurun_out/fl_alone.log
HRIEMO_FUSE_LN=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "train_step or determin or dropout_equals or seeded or golden" > gpurun_out/fl_parity.log 2>&1; echo "parity rc=$?" >> gpurun_out/fl_parity.log; tail -n 3 gpurun_out/fl_parity.log
for i in 1 2 3; do
  for v in 0 1; do
    HRIEMO_FUSE_LN=$v timeout -k 10 200 python bench.py --steps 40 --no-cpu-baseline --no-roofline 2> gpurun_out/fl_b$v$i.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('FUSE_LN=$v run $i:', d['ms_per_step'], d['ms_per_step_events']['median'])" || exit 4
  done
done
