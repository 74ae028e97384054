#!/usr/bin/env python3
"""GPU occupancy of one optimisation step from a rocprofv3 --kernel-trace run of bench.py (both streams): span of the
last complete step, union of the kernels' busy intervals (= span - idle), sum of kernel durations (> union where the two
streams overlap), the largest idle gaps, and kernel time per queue.  usage: step_overlap.py <dir with k_kernel_trace.csv> <steps>"""
import csv
import sys

d, steps = sys.argv[1], int(sys.argv[2])
tr = list(csv.DictReader(open(f"{d}/k_kernel_trace.csv")))
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
big = [i for i, r in enumerate(tr) if "pack4x4_batched" in r["Kernel_Name"] or "pack_all" in r["Kernel_Name"]]
marks = [i for i, r in enumerate(tr) if "p2p_pack_kernel" in r["Kernel_Name"] or "head_loss" in r["Kernel_Name"]]
# a step = from one occurrence of the step's first kernel to the next; take the second-to-last pair
firsts = [i for i in marks]
per = max(1, len(firsts) // steps)
a, b = firsts[-2 * per], firsts[-per]
seg = tr[a:b]
t0, t1 = int(seg[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in seg)
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
busy, cur_s, cur_e, gaps = 0, iv[0][0], iv[0][1], []
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, cur_e - t0))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e in iv)
print(f"launches {len(seg)}  span {(t1 - t0) / 1e3:.1f} us  busy(union) {busy / 1e3:.1f} us  idle {(t1 - t0 - busy) / 1e3:.1f} us  "
      f"sum of kernels {tot / 1e3:.1f} us  overlapped {(tot - busy) / 1e3:.1f} us")
gaps.sort(reverse=True)
print("idle gaps >= 2 us:", len([g for g in gaps if g[0] >= 2000]), " total", sum(g[0] for g in gaps if g[0] >= 2000) / 1e3, "us")
print("largest:", [(round(g[0] / 1e3, 1), round(g[1] / 1e3)) for g in gaps[:12]])
q = {}
for r in seg:
    k = r.get("Queue_Id", "?")
    q[k] = q.get(k, 0) + int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("kernel time per queue (us):", {k: round(v / 1e3, 1) for k, v in q.items()})
