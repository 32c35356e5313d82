"""Per-kernel PMC summary from rocprofv3 rocpd databases (one DB per --pmc pass).
python scripts/rocpd_pmc.py <db> [<db> ...]   -> markdown table: mean counter value per dispatch
(counter values are summed over the dimension instances of each dispatch first)."""
import sqlite3
import sys
from collections import defaultdict

KEEP = ("k_train", "k_sort", "k_march", "k_shade", "k_app", "k_mlp", "k_dense", "k_alpha", "k_finalize", "k_scan", "k_bwd", "k_pack", "k_scatter", "k_wgrad", "k_bin", "k_scene")
rows = defaultdict(dict)
dur = {}
for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    q = ("select kernel_name, counter_name, dispatch_id, sum(value), max(duration) from counters_collection "
         "group by kernel_name, counter_name, dispatch_id")
    acc = defaultdict(list)
    for name, ctr, disp, val, d in db.execute(q):
        short = name.split("(")[0].replace("lrf::", "").replace("void ", "")
        if not any(k in short for k in KEEP):
            continue
        acc[(short, ctr)].append(val)
        dur.setdefault(short, []).append(d)
    for (short, ctr), vals in acc.items():
        rows[short][ctr] = sum(vals) / len(vals)
ctrs = sorted({c for r in rows.values() for c in r})
print("| kernel | avg us (profiled) | " + " | ".join(ctrs) + " |")
print("|---|---|" + "---|" * len(ctrs))
for k, r in rows.items():
    print(f"| `{k}` | {sum(dur[k]) / len(dur[k]) / 1e3:.1f} | " + " | ".join(f"{r.get(c, float('nan')):.4g}" for c in ctrs) + " |")
