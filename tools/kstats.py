"""Per-kernel statistics from a rocprofv3 rocpd database (gpurun_out/<dir>/*_results.db), optionally the launch-by-launch
durations of the kernels whose name contains a pattern:  python tools/kstats.py <db> [pattern ...]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, count(*), avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, sum(end-start)/1e6 "
                 "from kernels group by name order by 6 desc").fetchall()
print(f"{'kernel':78s} {'calls':>5s} {'avg us':>10s} {'min us':>10s} {'max us':>10s} {'total ms':>9s}")
for r in rows[:int(20)]:
    print(f"{r[0][:78]:78s} {r[1]:5d} {r[2]:10.1f} {r[3]:10.1f} {r[4]:10.1f} {r[5]:9.1f}")
for pat in sys.argv[2:]:
    rows = c.execute("select (end-start)/1e3 from kernels where name like ? order by start", (f"%{pat}%",)).fetchall()
    print(pat, [round(r[0]) for r in rows])
