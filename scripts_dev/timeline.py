"""Device timeline of the graph-replayed step from a rocprofv3 --kernel-trace rocpd database: how much of the wall time
has 0 / 1 / >=2 kernels in flight, and where the idle gaps are.   usage: timeline.py <prof_dir> [steps_in_window]"""
import glob, os, sqlite3, sys
from collections import defaultdict

src = sys.argv[1]
f = glob.glob(os.path.join(src, "**", "*.db"), recursive=True)
d = sqlite3.connect(f[0])
cols = [r[1] for r in d.execute("pragma table_info(kernels)")]
print("kernels view columns:", cols)
qcol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
rows = list(d.execute(f"select name,start,end,{qcol or '0'},grid_x,workgroup_x from kernels order by start"))
blocks_of = {}
for r in rows:
    blocks_of[(r[0], r[1])] = r[4] // max(r[5], 1)
rows = [r[:4] for r in rows]
print("dispatches:", len(rows), "queues/streams:", sorted(set(r[3] for r in rows)))
# steady window: the last NSTEP steps; a step ends with its launch-boundary reduce (colreduce_batch_kernel)
NSTEP = int(sys.argv[2]) if len(sys.argv) > 2 else 10
marks = [r[2] for r in rows if r[0].startswith("colreduce_batch_kernel")]
t0, t1 = marks[-NSTEP - 1], marks[-1]
win = [r for r in rows if r[1] >= t0 and r[2] <= t1]
print(f"{NSTEP} steps: {(t1 - t0) / NSTEP / 1e6:.3f} ms/step between step-end marks")
span = win[-1][2] - win[0][1]
ev = []
for n, s, e, q in win:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth, last, hist = 0, ev[0][0], defaultdict(int)
for t, dlt in ev:
    hist[min(depth, 3)] += t - last
    last = t
    depth += dlt
print(f"window {span / 1e6:.2f} ms, {len(win)} dispatches")
for k in sorted(hist):
    print(f"  {k}{'+' if k == 3 else ''} kernels in flight: {hist[k] / 1e6:8.3f} ms  {100.0 * hist[k] / span:5.1f} %")
# idle gaps (depth 0) with the kernels on either side
gaps = []
depth, prev_end_name = 0, None
active_end = None
cur_end, cur_name = win[0][2], win[0][0]
for n, s, e, q in win[1:]:
    if s > cur_end:
        gaps.append((s - cur_end, cur_name, n))
    if e > cur_end:
        cur_end, cur_name = e, n
gaps.sort(reverse=True)
tot = sum(g[0] for g in gaps)
print(f"idle gaps: {len(gaps)}, total {tot / 1e6:.3f} ms; by size:")
for lo, hi in ((0, 1000), (1000, 2000), (2000, 4000), (4000, 8000), (8000, 16000), (16000, 10 ** 12)):
    sel = [g[0] for g in gaps if lo <= g[0] < hi]
    print(f"  {lo / 1e3:5.0f}-{hi / 1e3 if hi < 10 ** 11 else float('inf'):5.0f} us: {len(sel):5d} gaps, {sum(sel) / 1e6:.3f} ms")
short = lambda s: s.replace("void ", "")[:60]
print("largest idle gaps (us, after -> before):")
for g, a, b in gaps[:25]:
    print(f"  {g / 1e3:7.1f}  {short(a)}  ->  {short(b)}")
by = defaultdict(lambda: [0, 0])
for g, a, b in gaps:
    by[short(a).split("<")[0].split("(")[0]][0] += g
    by[short(a).split("<")[0].split("(")[0]][1] += 1
print("idle time by the kernel that ended before the gap:")
for k, (t, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:15]:
    print(f"  {t / 1e6:7.3f} ms in {c:5d} gaps after {k}")

# the last step, every idle gap >= 15 us with its time since the step's first kernel
last = [r for r in rows if r[1] >= marks[-2] and r[2] <= marks[-1]]
print(f"last step: {len(last)} dispatches, {(last[-1][2] - last[0][1]) / 1e6:.3f} ms from first start to last end; "
      f"gap before its first kernel {(last[0][1] - marks[-2]) / 1e3:.1f} us")
cur_end, cur_name = last[0][2], last[0][0]
for n, s_, e, q in last[1:]:
    if s_ - cur_end >= 15000:
        print(f"  t={(cur_end - last[0][1]) / 1e3:8.1f} us  idle {(s_ - cur_end) / 1e3:7.1f} us  {short(cur_name)} -> {short(n)}")
    if e > cur_end:
        cur_end, cur_name = e, n

# who runs ALONE: time with exactly one kernel in flight, by kernel (and how many blocks it has)
ev = []
for n, s_, e, q in win:
    ev.append((s_, 1, n, blocks_of[(n, s_)])); ev.append((e, -1, n, blocks_of[(n, s_)]))
ev.sort(key=lambda x: (x[0], x[1]))
active, lastt = {}, ev[0][0]
alone = defaultdict(lambda: [0, 0])
alone_small = 0
for t, dlt, n, b in ev:
    if len(active) == 1:
        (kn, kb), = active.keys()
        key = short(kn).split("(")[0][:48]
        alone[key][0] += t - lastt
        alone[key][1] = max(alone[key][1], kb)
        if kb < 256:
            alone_small += t - lastt
    lastt = t
    if dlt == 1:
        active[(n, b)] = active.get((n, b), 0) + 1
    else:
        active[(n, b)] -= 1
        if active[(n, b)] == 0:
            del active[(n, b)]
print(f"time with ONE kernel in flight, per step, by kernel (max blocks); with < 256 blocks: {alone_small / NSTEP / 1e6:.3f} ms/step")
for k, (t, b) in sorted(alone.items(), key=lambda kv: -kv[1][0])[:24]:
    print(f"  {t / NSTEP / 1e6:7.3f} ms  blocks<={b:6d}  {k}")

# per-stream busy time (a stream's kernels never overlap each other)
busy = defaultdict(int)
cnt = defaultdict(int)
for n, s_, e, q in win:
    busy[q] += e - s_
    cnt[q] += 1
print("per-stream busy time per step:")
for q in sorted(busy):
    print(f"  stream {q}: {busy[q] / NSTEP / 1e6:.3f} ms in {cnt[q] / NSTEP:.0f} dispatches")
# main-stream stalls: gaps on the busiest stream while the other stream is running (waiting at a join)
main = max(busy, key=busy.get)
mk = sorted([r for r in win if r[3] == main], key=lambda r: r[1])
stall = 0
big = []
for a, b in zip(mk, mk[1:]):
    g = b[1] - a[2]
    if g > 3000:
        stall += g
        big.append((g, a[0], b[0]))
big.sort(reverse=True)
print(f"gaps > 3 us on stream {main}: {stall / NSTEP / 1e6:.3f} ms/step; largest:")
for g, a, b in big[:12]:
    print(f"  {g / 1e3:7.1f} us  {short(a)} -> {short(b)}")
