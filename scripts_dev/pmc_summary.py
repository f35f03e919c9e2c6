"""Per-kernel-class sums of a rocprofv3 --pmc pass (rocpd database): usage pmc_summary.py <prof_dir> <out.json>"""
import glob, json, os, re, sqlite3, sys
from collections import defaultdict

src, out = sys.argv[1], sys.argv[2]
f = glob.glob(os.path.join(src, "**", "*.db"), recursive=True)
d = sqlite3.connect(f[0])
cols = [r[1] for r in d.execute("pragma table_info(counters_collection)")]
print("counters_collection columns:", cols)


def klass(name):
    m = re.search(r"gemm_kernel<(\d), (\d), (\d), (\d+), (\d+)", name)
    if m:
        lay = {"00": "nt", "01": "nn", "11": "tn"}[m.group(1) + m.group(2)]
        return f"gemm_{lay}_{m.group(4)}x{m.group(5)}"
    m = re.search(r"(attn_\w+_kernel)<(\d+), (\d+)", name)
    if m:
        return f"{m.group(1)}<{m.group(2)},{m.group(3)}>"
    return re.sub(r"[<(].*", "", name).replace("void ", "").strip()


tcol = [c for c in ("start", "start_timestamp") if c in cols]
ecol = [c for c in ("end", "end_timestamp") if c in cols]
dcol = "dispatch_id" if "dispatch_id" in cols else "id"
sel = f"select kernel_name,counter_name,value,{dcol}" + (f",{tcol[0]},{ecol[0]}" if tcol and ecol else "") + " from counters_collection"
agg = defaultdict(lambda: defaultdict(float))
seen = {}
for row in d.execute(sel):
    k = klass(row[0])
    agg[k][row[1]] += row[2]
    if len(row) > 4 and (k, row[3]) not in seen:
        seen[(k, row[3])] = 1
        agg[k]["_ns"] += row[5] - row[4]
        agg[k]["_n"] += 1
res = {}
for k, v in agg.items():
    if k.startswith("at::") or k.startswith("__amd"):
        continue
    res[k] = {kk: vv for kk, vv in v.items()}
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
hdr = sorted({c for v in res.values() for c in v if not c.startswith("_")})
print("class, launches, us/launch, " + ", ".join(hdr))
for k, v in sorted(res.items(), key=lambda kv: -kv[1].get("_ns", 0)):
    n = max(v.get("_n", 1), 1)
    print(f"{k}, {int(n)}, {v.get('_ns', 0) / n / 1e3:.1f}, " + ", ".join(f"{v.get(c, 0) / n:.4g}" for c in hdr))
