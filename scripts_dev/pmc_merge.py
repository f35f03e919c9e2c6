"""Merge the per-pass summaries of scripts_dev/prof_pmc.sh (mfma / stall / valu passes, one rocprofv3 --pmc run each) into the
JSON pmc_table.py prints: usage pmc_merge.py <mfma.json> <stall.json> <valu.json> <out.json>"""
import json, sys
res = {}
for tag, path in zip(("mfma", "stall", "valu"), sys.argv[1:4]):
    for k, v in json.load(open(path)).items():
        r = res.setdefault(k, {})
        for kk, vv in v.items():
            if kk in ("_ns", "_n"):
                if tag == "mfma":
                    r[kk] = vv
            elif kk == "SQ_WAVE_CYCLES":
                r[kk if tag == "stall" else "SQ_WAVE_CYCLES_" + tag] = vv
            else:
                r[kk] = vv
for r in res.values():      # the VALU share is taken against the wave cycles of ITS pass
    if "SQ_ACTIVE_INST_VALU" in r and r.get("SQ_WAVE_CYCLES_valu") and r.get("SQ_WAVE_CYCLES"):
        r["SQ_ACTIVE_INST_VALU"] *= r["SQ_WAVE_CYCLES"] / r["SQ_WAVE_CYCLES_valu"]
json.dump(res, open(sys.argv[4], "w"), indent=1, sort_keys=True)
