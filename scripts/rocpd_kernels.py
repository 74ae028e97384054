"""Average kernel durations out of a rocprofv3 rocpd database (the default output format of ROCm 7.2's rocprofv3):
python scripts/rocpd_kernels.py <results.db> [min launches]."""
import sqlite3
import sys
from collections import OrderedDict

db = sqlite3.connect(sys.argv[1])
mn = int(sys.argv[2]) if len(sys.argv) > 2 else 1
agg = OrderedDict()
for name, s, e, gx, gy, gz, wx in db.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels order by start"):
    k = (name[:86], gx // max(wx, 1), gy, gz)
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += (e - s) / 1e3
for k, a in agg.items():
    if a[0] >= mn:
        print(f"{k[0]:86s} wgs {k[1]:6d},{k[2]:4d},{k[3]:3d} n {a[0]:5d} avg {a[1] / a[0]:8.1f} us")
