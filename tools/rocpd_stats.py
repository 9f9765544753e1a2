"""Kernel statistics (as rocprofv3 --stats prints them) from a rocprofv3 rocpd database: python tools/rocpd_stats.py results.db out.csv"""
import csv, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) from kernels group by name order by 3 desc").fetchall()
total = sum(r[2] for r in rows)
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for n, c, t, a, mn, mx in rows:
        w.writerow([n, c, t, round(a, 1), round(100.0 * t / total, 2), mn, mx])
for n, c, t, a, mn, mx in rows[:22]:
    print(f"{100.0*t/total:6.2f}% {c:6d} calls {a/1e3:9.1f} us avg  {t/1e6:9.2f} ms  {n[:110]}")
