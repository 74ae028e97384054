#!/usr/bin/env python3
"""Per-step summary of a rocprofv3 --kernel-trace --stats run of bench.py: kernel totals per step and (with --timeline)
the launches of the last complete step in order.  usage: step_timeline.py <dir with k_kernel_stats.csv> <steps incl. warm-up> [--timeline [min_us]]"""
import csv
import re
import sys

d, steps = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(f"{d}/k_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel time {tot / 1e6 / steps:.3f} ms/step over {steps} steps")


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)
    return n[:86]


for r in rows[:int(sys.argv[5]) if len(sys.argv) > 5 else 34]:
    print(f"{short(r['Name']):86s} {int(r['Calls']) / steps:6.1f}/step {float(r['TotalDurationNs']) / 1e6 / steps:7.3f} ms "
          f"{float(r['AverageNs']) / 1e3:7.1f} us {float(r['Percentage']):5.1f}%")
if "--timeline" in sys.argv:
    mn = float(sys.argv[sys.argv.index("--timeline") + 1]) if len(sys.argv) > sys.argv.index("--timeline") + 1 else 10.0
    tr = list(csv.DictReader(open(f"{d}/k_kernel_trace.csv")))
    tr.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(tr) if "adam_kernel" in r["Kernel_Name"]]
    per = len(idx) // steps
    seg = tr[idx[-2 * per] + 1: idx[-per] + 1] if len(idx) >= 2 * per else tr
    t0 = int(seg[0]["Start_Timestamp"])
    for r in seg:
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if dur >= mn:
            print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {dur:8.1f} us wgs=({int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])},"
                  f"{r['Grid_Size_Y']},{r['Grid_Size_Z']}) {short(r['Kernel_Name'])}")
    print("span", (int(seg[-1]["End_Timestamp"]) - t0) / 1e3, "us,", len(seg), "launches")
