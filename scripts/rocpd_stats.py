"""Summarise a rocprofv3 rocpd database (kernel trace) into a per-kernel stats table:
python scripts/rocpd_stats.py <results.db> [out.md]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute(
    "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
    "max(vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(workgroup_x), max(grid_x) "
    "from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows) or 1
lines = ["| kernel | calls | total ms | avg us | min us | max us | % | vgpr | sgpr | lds B | scratch B | wg | grid |",
         "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
for r in rows:
    name = r[0][:70]
    lines.append(f"| `{name}` | {r[1]} | {r[2] / 1e6:.3f} | {r[3] / 1e3:.2f} | {r[4] / 1e3:.2f} | {r[5] / 1e3:.2f} | "
                 f"{100 * r[2] / tot:.1f} | {r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]} | {r[11]} |")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "a").write(out + "\n")
