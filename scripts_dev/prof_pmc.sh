#!/bin/bash
# GPU box: MFMA-pipe / wave-cycle counters per kernel class of the cfg-2 step (three --pmc passes, kernel tracing only) ->
# gpurun_out/pmc/pmc_mfma.{json,txt}
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
P=gpurun_out/pmc
rm -rf $P; mkdir -p $P
export EAGER=1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace -d $P/mfma -- python3 scripts_dev/replay_only.py 2 > $P/mfma.log 2>&1 || exit 3
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $P/stall -- python3 scripts_dev/replay_only.py 2 > $P/stall.log 2>&1 || exit 4
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace -d $P/valu -- python3 scripts_dev/replay_only.py 2 > $P/valu.log 2>&1 || exit 5
for p in mfma stall valu; do python3 scripts_dev/pmc_summary.py $P/$p $P/$p.json > /dev/null || exit 6; done
python3 scripts_dev/pmc_merge.py $P/mfma.json $P/stall.json $P/valu.json $P/pmc_mfma.json || exit 6
python3 scripts_dev/pmc_table.py $P/pmc_mfma.json > $P/pmc_mfma.txt || exit 7
rm -rf $P/mfma $P/stall $P/valu
head -30 $P/pmc_mfma.txt
