"""Kernel timeline of a rocprofv3 (rocpd SQLite) trace: name, queue, start offset, duration, gap to the previous kernel's end.
usage: python profiles/debug/timeline.py results.db [first] [count]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 80
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = db.execute(f"select name, start, end, {qcol or 0} from kernels order by start").fetchall()
rows = rows[first:first + count]
t0 = rows[0][1]
prev_end = rows[0][1]
for name, s, e, q in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"<.*", "", name)[:40]
    print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:7.1f}  q{q}  {name}")
    prev_end = max(prev_end, e)
