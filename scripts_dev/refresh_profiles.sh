#!/bin/bash
# Runs on the GPU box (gpurun): the bench line plus the rocprofv3 passes behind profiles/, exported as small summaries
# into gpurun_out/prof_summary (the raw rocpd databases exceed what gpurun merges back).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
P=gpurun_out/prof_raw
S=gpurun_out/prof_summary
rm -rf $P $S; mkdir -p $P $S
timeout -k 10 400 python3 bench.py > $S/bench.log 2>&1 || exit 2
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $P/default -- python3 bench.py --no-cpu-baseline --no-roofline > $P/default.log 2>&1 || exit 3
HRIEMO_TWO_STREAMS=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $P/serial -- python3 bench.py --no-graph --no-cpu-baseline --no-roofline > $P/serial.log 2>&1 || exit 4
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $P/fetch -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline > $P/fetch.log 2>&1 || exit 5
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d $P/write -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline > $P/write.log 2>&1 || exit 6
python3 scripts_dev/export_prof.py $P $S || exit 7
rm -rf $P
tail -1 $S/bench.log | cut -c1-400
