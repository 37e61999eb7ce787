"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace): per-kernel count / avg / min / max."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = list(cur.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) "
                        "from kernels group by name order by 6 desc"))
tot = sum(r[5] for r in rows)
print(f"{'kernel':72s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'total_ms':>9s} {'pct':>6s}")
for r in rows:
    print(f"{r[0][:72]:72s} {r[1]:6d} {r[2]/1e3:9.2f} {r[3]/1e3:9.2f} {r[4]/1e3:9.2f} {r[5]/1e6:9.2f} {100*r[5]/tot:6.1f}")
