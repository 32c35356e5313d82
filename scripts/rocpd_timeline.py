"""One iteration of a traced loop as a timeline: the kernels between the last two launches of a marker kernel, in start
order, with their offset from the iteration's start, duration, the idle gap in front of each, and which kernels overlap.
python scripts/rocpd_timeline.py <results.db> <marker substring> [iterations back=2]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
marker = sys.argv[2]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rows = db.execute("select start, end, name from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if marker in r[2]]
if len(marks) < back + 1:
    print("marker not found often enough:", len(marks)); sys.exit(0)
lo, hi = marks[-back - 1], marks[-back]
it = rows[lo:hi]
t0 = it[0][0]
span = max(r[1] for r in it) - t0
busy_end, busy = t0, 0
print(f"{len(it)} kernels, span {span / 1e3:.1f} us")
print("| # | start us | dur us | gap before us | kernel |\n|---|---|---|---|---|")
for i, (s, e, n) in enumerate(it):
    gap = s - busy_end
    if e > busy_end:
        busy += e - max(s, busy_end)
        busy_end = e
    print(f"| {i} | {(s - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {gap / 1e3:.1f} | `{n.split('(')[0][:80]}` |")
print(f"union busy {busy / 1e3:.1f} us of {span / 1e3:.1f} us = {100 * busy / span:.1f} %")
