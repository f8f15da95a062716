"""One steady-state train step out of a rocprofv3 (rocpd SQLite) kernel trace, in launch order: per kernel its duration, the gap
to the end of the previous kernel on the device, and the grid.  A step = the kernels from one `gte_fold_batch_kernel` (the last
launch of a step) to the next; the median-length step of the last third of the trace is printed, with the median over those steps
of every position's duration beside it.
usage: python profiles/rocpd_sequence.py results.db [txt-out]"""
import re
import sqlite3
import statistics
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
grid = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
wg = "workgroup_x" if "workgroup_x" in cols else ("workgroup_size_x" if "workgroup_size_x" in cols else None)
sel = "name, start, end" + (f", {grid}" if grid else ", 0") + (f", {wg}" if wg else ", 1")
rows = db.execute(f"select {sel} from kernels order by start").fetchall()


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\s+", " ", n.split("(")[0])[:64]


ends = [i for i, r in enumerate(rows) if "gte_fold_batch_kernel" in r[0]]
steps = [(a + 1, b + 1) for a, b in zip(ends[:-1], ends[1:])]
steps = steps[len(steps) * 2 // 3:]
sig = {}
for a, b in steps:                                     # the steps of the most common launch sequence
    sig.setdefault(tuple(short(r[0]) for r in rows[a:b]), []).append((a, b))
seq, same = max(sig.items(), key=lambda kv: len(kv[1]))
wall = [rows[b - 1][2] - rows[a - 1][2] for a, b in same]
med = sorted(range(len(same)), key=lambda i: wall[i])[len(same) // 2]
a, b = same[med]
out = [f"{len(same)} of {len(steps)} steps share this launch sequence ({len(seq)} launches); step wall (fold end to fold end) median "
       f"{statistics.median(wall) / 1e3:.1f} us; the step printed: {wall[med] / 1e3:.1f} us",
       f"{'kernel':64s} {'us':>8s} {'median us':>10s} {'gap us':>7s} {'workgroups':>10s}"]
tot = gap_tot = 0.0
for k in range(a, b):
    r = rows[k]
    dur = (r[2] - r[1]) / 1e3
    gap = (r[1] - max(x[2] for x in rows[max(0, k - 6):k])) / 1e3      # (the assemble kernel runs on a second stream)
    medk = statistics.median((rows[s + (k - a)][2] - rows[s + (k - a)][1]) / 1e3 for s, _ in same)
    nwg = (r[3] // max(r[4], 1)) if r[3] else 0
    out.append(f"{short(r[0]):64s} {dur:8.1f} {medk:10.1f} {gap:7.1f} {nwg:10d}")
    tot += dur
    gap_tot += max(gap, 0.0)
out.append(f"{'sum':64s} {tot:8.1f} {'':10s} {gap_tot:7.1f}")
text = "\n".join(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text + "\n")
print(text)
