"""Kernel summary of a rocprofv3 rocpd database (the default output format here):  python tools/rocpd_stats.py <db> <steps> [csv_out]"""
import csv
import sqlite3
import sys

db, steps = sqlite3.connect(sys.argv[1]), int(sys.argv[2])
rows = list(db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
if len(sys.argv) > 3:
    with open(sys.argv[3], "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_ALL)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], f"{r[3]:.1f}", f"{100 * r[2] / tot:.4f}", r[4], r[5]])
print(f"kernel time {tot / steps / 1e6:.2f} ms/step, {sum(r[1] for r in rows) / steps:.0f} launches/step")
for r in rows[:int(sys.argv[4]) if len(sys.argv) > 4 else 40]:
    print(f"{r[2] / steps / 1e6:8.2f} ms {r[1] / steps:7.1f}  {r[0][:110]}")
