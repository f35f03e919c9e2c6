#!/bin/bash
# GPU box: per-kernel totals of the ragged headline step, padded vs packed (scripts_dev/prof_packed.py), into gpurun_out/pp/
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/pp
for m in padded packed; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/pp/$m -- python3 scripts_dev/prof_packed.py $m 20 > gpurun_out/pp/$m.log 2>&1 || exit 3
  python3 - $m <<'PY' || exit 4
import csv, glob, sqlite3, sys
m = sys.argv[1]
d = sqlite3.connect(glob.glob(f"gpurun_out/pp/{m}/**/*.db", recursive=True)[0])
with open(f"gpurun_out/pp/{m}_kernel_stats.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
    for n, c, t, a, p in d.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        w.writerow([n, c, round(t, 1), round(a, 2), round(p, 3)])
PY
  rm -rf gpurun_out/pp/$m
  grep "ms/step" gpurun_out/pp/$m.log
done
