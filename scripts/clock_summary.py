"""profiles/r04_clock.json out of scripts/clock_settle.py's two runs: python scripts/clock_summary.py <dir>
(<dir>/plain.jsonl, <dir>/profiled.jsonl, <dir>/pmc/**/*counter_collection.csv)."""
import csv, json, os, sys
src = sys.argv[1]
plain = [json.loads(l) for l in open(os.path.join(src, "plain.jsonl")) if l.startswith("{")]
prof = [json.loads(l) for l in open(os.path.join(src, "profiled.jsonl")) if l.startswith("{")]
path = None
for root, _, files in os.walk(os.path.join(src, "pmc")):
    for f in files:
        if f.endswith("counter_collection.csv"):
            path = os.path.join(root, f)
rows = [r for r in csv.DictReader(open(path)) if "conv3x3" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
out = {"note": "in-kernel clock = delta s_memtime / delta s_memrealtime x 100 MHz, median over the workgroups of ONE launch "
               "(S2S_CONV_DBG=64); grbm clock = GRBM_GUI_ACTIVE / 8 XCDs / that dispatch's duration.  'plain': the last of "
               "N back-to-back launches without a profiler (the chip under the kernel's own sustained load); 'profiled': "
               "the same script under rocprofv3 --pmc GRBM_GUI_ACTIVE, both readings for the same dispatch.",
       "shapes": []}
for a, b in zip(plain, prof):
    k = b["conv_dispatch_index_of_last_launch"] - 1
    r = rows[k]
    dur_ns = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    grbm = float(r["Counter_Value"])
    # the mean over the shape's profiled launches too (single dispatches are noisy)
    lo = k + 1 - b["launches"]
    g = [float(x["Counter_Value"]) / 8.0 / (float(x["End_Timestamp"]) - float(x["Start_Timestamp"])) * 1e3 for x in rows[lo:k + 1]]
    out["shapes"].append({"shape": a["shape"], "operands": a["operands"],
                          "plain": {"launches": a["launches"], "seconds_of_load": a["seconds_of_load"], "tflops": a["tflops"],
                                    "memtime_mhz": a["memtime_mhz_median"], "p10": a["memtime_mhz_p10"], "p90": a["memtime_mhz_p90"]},
                          "profiled": {"launches": b["launches"], "memtime_mhz_same_dispatch": b["memtime_mhz_median"],
                                       "grbm_mhz_same_dispatch": round(grbm / 8.0 / dur_ns * 1e3, 0),
                                       "grbm_mhz_mean_over_launches": round(sum(g) / len(g), 0),
                                       "dispatch_us": round(dur_ns / 1e3, 1), "kernel": r["Kernel_Name"][:60]}})
json.dump(out, open("profiles/r04_clock.json", "w"), indent=1)
print(json.dumps(out, indent=1))
