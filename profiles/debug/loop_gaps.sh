#!/bin/bash
# where the train loop's time goes beyond its kernels: per-step interval / busy / idle from a kernel trace, and the largest gaps
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/gaps; mkdir -p $O
STEP_ONLY="--no-cpu-baseline --no-gather-probe --no-secondary --no-cfg3 --no-replay --no-split-probe --no-inference --val-graph 0"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/t -o t -- python3 $R/bench.py --long-run-seconds 0.2 $STEP_ONLY > $O/log.txt 2>&1
DB=$(ls $O/t/*.db | head -1)
python3 $R/profiles/debug/step_gaps.py $DB | tail -12
python3 - $DB <<'PY'
import sqlite3, sys, numpy as np, collections
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
# gaps between consecutive kernels (any stream) in the steady state: which kernel FOLLOWS a gap > 3 us
n = len(rows); lo = n // 2; hi = n - 200
gaps = collections.defaultdict(list)
cur_end = rows[lo][2]
for i in range(lo + 1, hi):
    name, s, e = rows[i]
    if s > cur_end: gaps[name[:60]].append((s - cur_end) / 1e3)
    cur_end = max(cur_end, e)
steps = sum(1 for r in rows[lo:hi] if "gte_fold_batch_kernel" in r[0])
print("idle before kernel (us per step), steady state,", steps, "steps")
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:16]:
    print(f"  {sum(v) / steps:7.2f} us/step  n={len(v):5d}  mean {np.mean(v):6.2f}  max {max(v):8.1f}  {k}")
PY
rm -rf $O/t
