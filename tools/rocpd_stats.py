"""Per-kernel totals from a rocprofv3 rocpd database (kernel-trace): python tools/rocpd_stats.py <results.db> [steps]"""
import sqlite3, sys, re
con = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kt = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
name = 'display_name' if 'display_name' in cols else ('kernel_name' if 'kernel_name' in cols else cols[-1])
rows = cur.execute(f"select s.{name}, count(*), sum(d.end-d.start), avg(d.end-d.start) from {kt} d join {ks} s on d.kernel_id=s.id group by s.{name} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print("total kernel time %.3f ms%s" % (tot / 1e6, " (%.3f ms per step)" % (tot / 1e6 / steps) if steps != 1 else ""))
for n, c, t, a in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 28]:
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"\(.*", "", n)[:90]
    print("%6.1f%% %8.3f ms %7d x %9.2f us  %s" % (100 * t / tot, t / 1e6 / steps, c / steps if steps != 1 else c, a / 1e3, n))
