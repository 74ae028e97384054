"""Per-kernel totals out of a rocprofv3 rocpd database: python scripts/rocpd_stats.py <results.db> <steps> [csv out].
Prints calls / average / total per step, sorted by total (the --stats table of the CSV output format)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = {}
for name, s, e in db.execute("select name, start, end from kernels"):
    a = agg.setdefault(name, [0, 0.0, 1e30, 0.0])
    d = (e - s)
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values())
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
if len(sys.argv) > 3:
    with open(sys.argv[3], "w") as f:
        f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"\n')
        for k, a in rows:
            f.write(f'"{k}",{a[0]},{int(a[1])},{a[1] / a[0]:.1f},{100 * a[1] / tot:.2f},{int(a[2])},{int(a[3])}\n')
print(f"total kernel time per step {tot / steps / 1e6:.3f} ms, launches per step {sum(a[0] for a in agg.values()) / steps:.1f}")
for k, a in rows:
    print(f"{k[:96]:96s} {a[0] / steps:6.1f}/step avg {a[1] / a[0] / 1e3:8.1f} us  {a[1] / steps / 1e3:8.1f} us/step {100 * a[1] / tot:5.1f}%")
