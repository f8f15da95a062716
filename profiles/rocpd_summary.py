"""Per-kernel summary of a rocprofv3 (rocpd SQLite) kernel trace.
usage: python profiles/rocpd_summary.py results.db [csv-out]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                  "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
out = ["pct,calls,total_us,avg_us,min_us,max_us,kernel"]
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r[0])
    name = re.sub(r"\s+", " ", name)[:150].replace(",", ";")
    out.append(f"{r[2] / tot * 100:.2f},{r[1]},{r[2]:.1f},{r[3]:.2f},{r[4]:.2f},{r[5]:.2f},{name}")
text = "\n".join(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text + "\n")
print("\n".join(out[:32]))
