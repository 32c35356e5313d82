"""GPU busy fraction over the tail of a rocprofv3 kernel trace (rocpd database):
python scripts/rocpd_busy.py <results.db> [tail_fraction=0.2]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.2
rows = db.execute("select start, end, name from kernels order by start").fetchall()
t0, t1 = rows[0][0], max(r[1] for r in rows)
cut = t1 - (t1 - t0) * frac
tail = [r for r in rows if r[0] >= cut]
busy = sum(r[1] - r[0] for r in tail)
span = max(r[1] for r in tail) - tail[0][0]
print(f"kernels in tail: {len(tail)}  span {span / 1e6:.2f} ms  busy {busy / 1e6:.2f} ms  = {100 * busy / span:.1f} % GPU busy")
agg = {}
for s, e, n in tail:
    k = n.split("(")[0][:60]
    a = agg.setdefault(k, [0, 0]); a[0] += 1; a[1] += e - s
for k, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {d / 1e6:8.3f} ms  {c:5d}x  {k}")
