"""Step-to-step intervals and idle time from a rocprofv3 kernel trace: one marker kernel per step (default: the fold launch).
usage: python profiles/debug/step_gaps.py results.db [marker substring]"""
import sqlite3, sys
import numpy as np
db = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else "gte_fold_batch_kernel"
rows = db.execute("select name, start, end from kernels order by start").fetchall()
st = np.array([r[1] for r in rows], dtype=np.float64); en = np.array([r[2] for r in rows], dtype=np.float64)
idx = [i for i, r in enumerate(rows) if marker in r[0]]
print(f"{len(rows)} kernels, {len(idx)} steps (marker {marker})")
for a, b in zip(idx[:-1], idx[1:]):
    pass
iv = np.diff(st[idx]) / 1e3
busy = []
for a, b in zip(idx[:-1], idx[1:]):
    s, e = st[a + 1:b + 1], en[a + 1:b + 1]
    # union length of the kernel intervals of this step (two streams overlap)
    order = np.argsort(s); cur_s, cur_e, tot = None, None, 0.0
    for j in order:
        if cur_s is None or s[j] > cur_e:
            if cur_s is not None: tot += cur_e - cur_s
            cur_s, cur_e = s[j], e[j]
        else:
            cur_e = max(cur_e, e[j])
    if cur_s is not None: tot += cur_e - cur_s
    busy.append(tot / 1e3)
busy = np.array(busy)
for lo in range(0, len(iv), 25):
    hi = min(lo + 25, len(iv))
    print(f"steps {lo:4d}-{hi:4d}: interval {np.median(iv[lo:hi]):7.1f} us (max {iv[lo:hi].max():8.1f})  busy {np.median(busy[lo:hi]):7.1f} us  idle {np.median(iv[lo:hi] - busy[lo:hi]):6.1f}")
