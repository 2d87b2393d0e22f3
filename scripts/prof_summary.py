"""Per-kernel summary of a rocprofv3 --kernel-trace run (the sqlite file it writes)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rows = db.execute("select name, count(*), sum(end-start)/1e3 as tot, avg(end-start)/1e3, grid_x, workgroup_x from kernels group by name, grid_x order by tot desc").fetchall()
T = sum(r[2] for r in rows)
for r in rows[:top]:
    print("  %-64s grid %-7s n %6d total %9.1f us (%4.1f%%) avg %7.2f us" % (r[0][:64], r[4] // max(1, r[5]), r[1], r[2], 100 * r[2] / T, r[3]))
span = db.execute("select min(start), max(end) from kernels").fetchone()
print("  kernel time total %.1f ms" % (T / 1e3))
