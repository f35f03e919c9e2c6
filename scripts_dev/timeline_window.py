"""Dispatches of the last replayed step between two offsets (us) from its first kernel: start, end, duration, stream, blocks, name.
usage: timeline_window.py <prof_dir> <from_us> <to_us>"""
import glob, os, sqlite3, sys
src, t_from, t_to = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
f = glob.glob(os.path.join(src, "**", "*.db"), recursive=True)
d = sqlite3.connect(f[0])
rows = list(d.execute("select name,start,end,stream_id,grid_x,workgroup_x from kernels order by start"))
marks = [r[2] for r in rows if r[0].startswith("colreduce_batch_kernel")]
k = int(sys.argv[4]) if len(sys.argv) > 4 else 1
last = [r for r in rows if r[1] >= marks[-k - 1] and r[2] <= marks[-k]]
t0 = last[0][1]
for n, s, e, q, g, w in last:
    a, b = (s - t0) / 1e3, (e - t0) / 1e3
    if b >= t_from and a <= t_to:
        print(f"{a:9.1f} {b:9.1f} {b - a:7.1f} us  stream {q}  blocks {g // max(w, 1):6d}  {n[:90]}")
