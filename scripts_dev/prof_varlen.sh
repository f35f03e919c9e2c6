#!/bin/bash
# GPU box: kernel statistics of the ragged step, padded and packed, kernels serial (one stream, eager) -> gpurun_out/varlen_prof
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
P=gpurun_out/vprof_raw
S=gpurun_out/varlen_prof
rm -rf $P $S; mkdir -p $P $S
export HRIEMO_TWO_STREAMS=0 EAGER=1 RAGGED=1
VARLEN=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $P/default -- python3 scripts_dev/replay_only.py 20 > $S/padded.log 2>&1 || exit 3
VARLEN=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $P/serial -- python3 scripts_dev/replay_only.py 20 > $S/packed.log 2>&1 || exit 4
python3 scripts_dev/export_prof.py $P $S || exit 7
mv $S/bench_kernel_stats.csv $S/padded_kernel_stats.csv; mv $S/bench_serial_kernel_stats.csv $S/packed_kernel_stats.csv
rm -rf $P
tail -1 $S/padded.log $S/packed.log
