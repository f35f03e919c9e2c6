#!/bin/bash
# GPU box: the round's closing run -- full GPU suite, both soaks with their configuration lines, the bench line
set -o pipefail
mkdir -p gpurun_out
(timeout -k 10 700 python -m pytest tests -x -q -m gpu > gpurun_out/r4_suite_final.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_suite_final.log); tail -n 4 gpurun_out/r4_suite_final.log
timeout -k 10 200 python scripts_dev/soak_step.py 150 64 > gpurun_out/r4_soak.log 2>&1 || exit 3; tail -n 2 gpurun_out/r4_soak.log
VARLEN=1 timeout -k 10 200 python scripts_dev/soak_step.py 150 64 ragged > gpurun_out/r4_soak_packed.log 2>&1 || exit 4; tail -n 2 gpurun_out/r4_soak_packed.log
