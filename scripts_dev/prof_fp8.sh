#!/bin/bash
# GPU box: kernel statistics of the cfg-5 step, bf16 and MX-fp8 forward GEMMs, kernels serial (one stream, eager) -> gpurun_out/fp8_prof
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
P=gpurun_out/fprof_raw
S=gpurun_out/fp8_prof
rm -rf $P $S; mkdir -p $P $S
export HRIEMO_TWO_STREAMS=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $P/default -- python3 bench.py --workload cfg5 --no-graph --no-cpu-baseline --no-roofline --steps 10 --warmup 2 > $S/bf16.log 2>&1 || exit 3
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $P/serial -- python3 bench.py --workload cfg5_fp8 --no-graph --no-cpu-baseline --no-roofline --steps 10 --warmup 2 > $S/fp8.log 2>&1 || exit 4
python3 scripts_dev/export_prof.py $P $S || exit 7
mv $S/bench_kernel_stats.csv $S/bf16_kernel_stats.csv; mv $S/bench_serial_kernel_stats.csv $S/fp8_kernel_stats.csv
rm -rf $P
