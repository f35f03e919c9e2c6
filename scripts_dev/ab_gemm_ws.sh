#!/bin/bash
# GPU box: config 9 (loader / consumer waves) -- GEMM tests, sweep against configs 1 / 2, behaviour with CUs held by another
# kernel, step A/B: default (1) vs static walk (9 = bit 3) vs queue kernels (3 = bit 1)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 120 python scripts_dev/ab_gemm_flags.py 2>&1 | grep "^exact" || exit 1
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm" > gpurun_out/ws_tests.log 2>&1; rc=$?; tail -n 2 gpurun_out/ws_tests.log; [ $rc = 0 ] || exit 2
CFGS=1,2,9 timeout -k 10 300 python scripts_dev/bench_gemm.py > gpurun_out/ws_sweep.log 2>&1 || { tail -5 gpurun_out/ws_sweep.log; exit 3; }
grep -v amdgpu.ids gpurun_out/ws_sweep.log | tail -n 31
timeout -k 10 200 python scripts_dev/bench_hog.py 2>&1 | grep "^NT" || exit 5
timeout -k 10 300 python scripts_dev/ab_step_flags.py 1 9 3 2>&1 | grep "^flags" || exit 4
