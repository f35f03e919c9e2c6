"""Turn rocprofv3's rocpd databases of the bench runs into the small summaries committed under profiles/:
usage: export_prof.py <prof_dir> <out_dir>   (prof_dir holds default/ serial/ fetch/ write/ sub-directories)"""
import csv, glob, json, os, re, sqlite3, sys
from collections import defaultdict

src, out = sys.argv[1], sys.argv[2]
os.makedirs(out, exist_ok=True)


def db(name):
    f = glob.glob(os.path.join(src, name, "**", "*.db"), recursive=True)
    return sqlite3.connect(f[0]) if f else None


def stats(name, path):
    d = db(name)
    if d is None:
        return
    rows = list(d.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    mm = {n: (mn, mx) for n, mn, mx in d.execute("select name,min(duration),max(duration) from kernels group by name")}
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for n, c, t, a, p in rows:
            w.writerow([n, c, int(t * 1000), round(a * 1000, 1), round(p, 3), mm.get(n, (0, 0))[0], mm.get(n, (0, 0))[1]])


def klass(name):
    m = re.search(r"gemm(?:_ws)?_kernel<(\d), (\d),", name)          # template kernels and the loader / consumer kernel
    if m:
        return {"00": "gemm_bf16_nt", "01": "gemm_bf16_nn", "11": "gemm_bf16_tn"}[m.group(1) + m.group(2)]
    return re.sub(r"\(.*", "", name).replace("void ", "").strip()


def collect(name, counter):
    d = db(name)
    tot, cnt = defaultdict(float), defaultdict(int)
    if d is not None:
        for n, v in d.execute("select kernel_name,value from counters_collection where counter_name=?", (counter,)):
            k = klass(n)
            tot[k] += v * 1024.0
            cnt[k] += 1
    return tot, cnt


stats("default", os.path.join(out, "bench_kernel_stats.csv"))
stats("serial", os.path.join(out, "bench_serial_kernel_stats.csv"))
ft, fc = collect("fetch", "FETCH_SIZE")
wt, wc = collect("write", "WRITE_SIZE")
res = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py --steps 2 --warmup 1 --no-graph "
               "(4 steps incl. the eager sizing step); FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md HBM), counters in KiB",
       "kernels": {}}
for k in sorted(ft, key=lambda k: -ft[k]):
    if fc[k] and not k.startswith("__amd") and "at::" not in k:
        res["kernels"][k] = {"launches": fc[k], "fetch_bytes_per_launch": 2.0 * ft[k] / fc[k],
                             "write_bytes_per_launch": (wt[k] / wc[k]) if wc.get(k) else None}
json.dump(res, open(os.path.join(out, "hbm_traffic.json"), "w"), indent=1)
print("exported", sorted(os.listdir(out)))
