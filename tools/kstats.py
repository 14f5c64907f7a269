"""Per-kernel totals from a rocprofv3 results database (rocpd sqlite)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                  "from kernels group by name order by 3 desc").fetchall()
print("%-70s %7s %12s %10s %10s %10s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us"))
for r in rows:
    print("%-70s %7d %12.1f %10.1f %10.1f %10.1f" % (r[0][:70], r[1], r[2], r[3], r[4], r[5]))
