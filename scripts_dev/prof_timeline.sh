#!/bin/bash
# GPU box: device timeline of the replayed cfg-2 step (kernels in flight, idle gaps) -> gpurun_out/timeline/step_timeline.txt,
# every dispatch of the last step's first 450 us -> step_start.txt, and of its gate / decoder chain -> step_tail.txt
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
P=gpurun_out/timeline
rm -rf $P; mkdir -p $P
timeout -k 10 300 rocprofv3 --kernel-trace -d $P/raw -- python3 scripts_dev/replay_only.py 30 > $P/run.log 2>&1 || exit 3
python3 scripts_dev/timeline.py $P/raw 10 > $P/step_timeline.txt 2>&1 || exit 4
python3 scripts_dev/timeline_window.py $P/raw 0 450 > $P/step_start.txt 2>&1 || exit 5
python3 scripts_dev/timeline_window.py $P/raw 2000 4000 > $P/step_tail.txt 2>&1 || exit 6
rm -rf $P/raw
wc -l $P/step_tail.txt
