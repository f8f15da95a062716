"""avg duration of one kernel (substring match) in consecutive groups of a rocpd trace: python kernel_avgs.py db substr ngroups"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select start, end from kernels where name like ? order by start", (f"%{sys.argv[2]}%",)).fetchall()
g = int(sys.argv[3]); per = len(rows) // g
for i in range(g):
    part = rows[i * per:(i + 1) * per]
    d = sorted((e - s) / 1e3 for s, e in part)
    print(f"group {i}: {len(part)} launches, median {d[len(d)//2]:.1f} us, min {d[0]:.1f}, max {d[-1]:.1f}")
