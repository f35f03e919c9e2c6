"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into profiles/hbm_traffic.json.
usage: aggregate_pmc.py <dir_fetch> <dir_write> <out.json>
Counters are in KiB; FETCH_SIZE is doubled (gfx950 correction, MI355X_MICROARCH.md, HBM section)."""
import csv, glob, json, os, re, sys
from collections import defaultdict

def klass(name):
    m = re.search(r"gemm_kernel<(\d), (\d),", name)
    if m:
        return {"00": "gemm_bf16_nt", "01": "gemm_bf16_nn", "11": "gemm_bf16_tn"}[m.group(1) + m.group(2)]
    m = re.match(r"(?:void )?(\w+)", name)
    return re.sub(r"\(.*", "", name).strip() if not m else re.sub(r"\(.*", "", name).replace("void ", "").strip()

def collect(d, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = klass(r["Kernel_Name"])
            tot[k] += float(r["Counter_Value"]) * 1024.0
            cnt[k] += 1
    return tot, cnt

fd, wd, out = sys.argv[1:4]
ft, fc = collect(fd, "FETCH_SIZE")
wt, wc = collect(wd, "WRITE_SIZE")
res = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py --steps 2 --warmup 1 --no-graph "
               "(4 steps incl. the eager sizing step); FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md HBM), counters in KiB",
       "kernels": {}}
for k in sorted(ft, key=lambda k: -ft[k]):
    if fc[k] == 0:
        continue
    res["kernels"][k] = {"launches": fc[k], "fetch_bytes_per_launch": 2.0 * ft[k] / fc[k],
                         "write_bytes_per_launch": (wt.get(k, 0.0) / wc[k]) if wc.get(k) else None}
json.dump(res, open(out, "w"), indent=1)
print("wrote", out, "with", len(res["kernels"]), "kernels")
