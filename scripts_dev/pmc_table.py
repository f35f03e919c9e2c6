"""Table of scripts_dev/pmc_summary.py's JSON in the form of profiles/r0N_pmc_mfma.txt:  usage pmc_table.py <summary.json>"""
import json, sys
res = json.load(open(sys.argv[1]))
print("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY")
print("#   SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -- python3 scripts_dev/replay_only.py 2   (EAGER=1, kernels serialized by the profiler)")
print("# mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs * 1024 SIMDs); wave-cycle split: parked (s_waitcnt/barrier) / issue-stalled / issuing")
print("# clock_GHz = GRBM_GUI_ACTIVE/8 / duration.")
print(f"{'kernel class':34s} {'launches':>8s} {'us':>7s} {'GHz':>5s} {'mfma%':>6s} {'parked%':>8s} {'stall%':>7s} {'issue%':>7s} {'valu%':>6s}")
for k, v in sorted(res.items(), key=lambda kv: -kv[1].get("_ns", 0))[:28]:
    n = max(v.get("_n", 1), 1)
    us = v.get("_ns", 0) / n / 1e3
    gui = v.get("GRBM_GUI_ACTIVE", 0) / n / 8
    wave = max(v.get("SQ_WAVE_CYCLES", 0), 1)
    mfma = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / n / max(gui * 1024, 1) * 100
    parked = v.get("SQ_WAIT_ANY", 0) / wave * 100
    stall = v.get("SQ_WAIT_INST_ANY", 0) / wave * 100
    issue = v.get("SQ_ACTIVE_INST_ANY", 0) / wave * 100
    valu = v.get("SQ_ACTIVE_INST_VALU", 0) / wave * 100
    print(f"{k[:34]:34s} {int(n):8d} {us:7.1f} {gui / max(us, 1e-9) / 1e3:5.2f} {mfma:6.1f} {parked:8.1f} {stall:7.1f} {issue:7.1f} {valu:6.1f}")
