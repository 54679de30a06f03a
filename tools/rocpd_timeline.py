"""Timeline of the last step's kernels from a rocprofv3 rocpd database (kernel-trace): start, end, duration, queue, name -- to
see what overlaps what.  python tools/rocpd_timeline.py <results.db> [anchor substring = adam_kernel] [max rows]"""
import sqlite3, sys, re
con = sqlite3.connect(sys.argv[1])
anchor = sys.argv[2] if len(sys.argv) > 2 else "adam_kernel"
cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kt = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
name = 'display_name' if 'display_name' in cols else ('kernel_name' if 'kernel_name' in cols else cols[-1])
dcols = [r[1] for r in cur.execute(f"pragma table_info({kt})")]
q = 'queue_id' if 'queue_id' in dcols else ('stream_id' if 'stream_id' in dcols else None)
rows = cur.execute(f"select d.start, d.end, {('d.' + q) if q else '0'}, s.{name} from {kt} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
ends = [i for i, r in enumerate(rows) if anchor in r[3]]
if len(ends) < 2:
    sys.exit("fewer than two anchors (%s)" % anchor)
lo, hi = ends[-2] + 1, ends[-1] + 1
t0 = rows[lo][0]
prev_end = t0
for s, e, qq, n in rows[lo:hi][:int(sys.argv[3]) if len(sys.argv) > 3 else 400]:
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"\(.*", "", n)[:70]
    print("%9.1f %9.1f  %7.1f us  q%-3s %s%s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, qq, "| " if s < prev_end - 500 else "", n))
    prev_end = max(prev_end, e)
print("step: %.1f us" % ((rows[hi - 1][1] - t0) / 1e3))
