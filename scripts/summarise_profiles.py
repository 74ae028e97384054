"""Turn gpurun_out/prof_<tag>/ (scripts/profile_round.sh) into the summaries committed under profiles/:
<tag>_bench.json, <tag>_bench_under_rocprof.json, <tag>_kernel_stats.csv, <tag>_hbm_traffic.json (+ the copy
bench.py reads, hbm_traffic_current.json).  HBM bytes per launch follow MI355X_MICROARCH.md's HBM section:
separate --pmc passes for FETCH_SIZE and WRITE_SIZE (KB units), FETCH_SIZE doubled for gfx950's wide reads."""
import csv, json, os, shutil, sys
from collections import defaultdict

tag = sys.argv[1]
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)
shutil.copy(f"{src}/bench.json", f"profiles/{tag}_bench.json")
shutil.copy(f"{src}/bench_under_rocprof.json", f"profiles/{tag}_bench_under_rocprof.json")
shutil.copy(f"{src}/stats/k_kernel_stats.csv", f"profiles/{tag}_kernel_stats.csv")
for a, b in (("p2p_stats/k_kernel_stats.csv", "pix2pix_kernel_stats.csv"), ("p2p_bench_under_rocprof.json", "pix2pix_bench_under_rocprof.json"),
             ("stats_overlap/k_kernel_stats.csv", "kernel_stats_two_streams.csv")):
    if os.path.exists(f"{src}/{a}"):
        shutil.copy(f"{src}/{a}", f"profiles/{tag}_{b}")


def per_kernel(path, counter):
    acc = defaultdict(list)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == counter:
            acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return acc


fetch = per_kernel(f"{src}/pmc_fetch/f_counter_collection.csv", "FETCH_SIZE")
write = per_kernel(f"{src}/pmc_write/w_counter_collection.csv", "WRITE_SIZE")
def _has(needle, name):
    return any(n in name for n in needle) if isinstance(needle, tuple) else needle in name


groups = {"conv3x3_mfma (fwd+dgrad: conv3x3_dma16_kernel + conv3x3_stage_kernel + conv3x3_pers16_kernel)": ("conv3x3_dma16_kernel", "conv3x3_pers16_kernel", "conv3x3_stage_kernel"),
          "conv3x3_wgrad (conv3x3_wgrad_dma_kernel)": "conv3x3_wgrad_dma_kernel",
          "bn_relu_apply": "bn_relu_apply_kernel", "bn_relu_bwd_reduce_flat": "bn_relu_bwd_reduce_flat_kernel",
          "bn_relu_bwd_apply_flat": "bn_relu_bwd_apply_flat_kernel", "upsample2x_fwd": "upsample2x_fwd_kernel",
          "upsample2x_bwd": "upsample2x_bwd_", "adam": "adam_kernel", "head_loss": "head_loss_lanes_kernel"}
out = {}
for label, needle in groups.items():
    f = [v for k, vs in fetch.items() if _has(needle, k) for v in vs]
    w = [v for k, vs in write.items() if _has(needle, k) for v in vs]
    if not f or not w:
        continue
    fk, wk = sum(f) / len(f), sum(w) / len(w)
    out[label] = {"launches_sampled": len(f), "FETCH_SIZE_KB_avg": round(fk, 1), "WRITE_SIZE_KB_avg": round(wk, 1),
                  "hbm_bytes_per_launch_raw": int((fk + wk) * 1024),
                  "hbm_bytes_per_launch_corrected": int((2 * fk + wk) * 1024),
                  "note": "gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM); "
                          "corrected = 2*FETCH + WRITE"}
json.dump(out, open(f"profiles/{tag}_hbm_traffic.json", "w"), indent=1)
json.dump(out, open("profiles/hbm_traffic_current.json", "w"), indent=1)
for k, v in out.items():
    print(f"{k:55s} {v['hbm_bytes_per_launch_corrected']/1e6:9.1f} MB/launch over {v['launches_sampled']} launches")

# HBM rate per kernel from the counters: corrected bytes per launch / average launch time of the kernel-stats pass
rates = {}
stats = list(csv.DictReader(open(f"{src}/stats/k_kernel_stats.csv")))
for label, needle in groups.items():
    if label not in out:
        continue
    tot = sum(float(r["TotalDurationNs"]) for r in stats if _has(needle, r["Name"]))
    n = sum(int(r["Calls"]) for r in stats if _has(needle, r["Name"]))
    if n:
        rates[label] = {"hbm_bytes_per_launch_corrected": out[label]["hbm_bytes_per_launch_corrected"],
                        "avg_launch_us": round(tot / n / 1e3, 2),
                        "hbm_gb_per_s_from_counters": round(out[label]["hbm_bytes_per_launch_corrected"] / (tot / n), 1)}
json.dump(rates, open(f"profiles/{tag}_hbm_rates.json", "w"), indent=1)

# ---- the pix2pix G + D step (row a13): same counters, its own kernels ----
if os.path.exists(f"{src}/p2p_pmc_fetch/f_counter_collection.csv"):
    fetch = per_kernel(f"{src}/p2p_pmc_fetch/f_counter_collection.csv", "FETCH_SIZE")
    write = per_kernel(f"{src}/p2p_pmc_write/w_counter_collection.csv", "WRITE_SIZE")
    stats = list(csv.DictReader(open(f"{src}/p2p_stats/k_kernel_stats.csv")))
    groups = {"convkxk (4x4 forward / data gradient / transposed)": "convkxk_dma16_kernel",
              "convsm (inner levels: conv + norm in one launch)": "convsm_kernel",
              "conv2x2_wgrad_small (inner levels, no slabs)": "conv2x2_wgrad_small_kernel",
              "conv2x2_wgrad (4x4 stride-2 weight gradient)": "conv2x2_wgrad_dma_kernel",
              "convkxk_wgrad_rows (4x4 stride-1 weight gradient)": "convkxk_wgrad_rows_kernel",
              "wgrad_fold4x4": "wgrad_fold4x4_kernel", "instnorm reduce": "in_reduce_kernel",
              "instnorm apply": "in_lrelu_apply_kernel", "instnorm bwd apply": "in_lrelu_bwd_apply_kernel",
              "adam": "adam_kernel", "pack4x4": "pack4x4_batched_kernel", "act_bwd": "p2p_act_bwd_kernel"}
    out = {}
    for label, needle in groups.items():
        f = [v for k, vs in fetch.items() if needle in k for v in vs]
        w = [v for k, vs in write.items() if needle in k for v in vs]
        tot = sum(float(r["TotalDurationNs"]) for r in stats if needle in r["Name"])
        n = sum(int(r["Calls"]) for r in stats if needle in r["Name"])
        if not f or not w or not n:
            continue
        fk, wk = sum(f) / len(f), sum(w) / len(w)
        out[label] = {"launches_sampled": len(f), "FETCH_SIZE_KB_avg": round(fk, 1), "WRITE_SIZE_KB_avg": round(wk, 1),
                      "hbm_bytes_per_launch_corrected": int((2 * fk + wk) * 1024),
                      "avg_launch_us": round(tot / n / 1e3, 2),
                      "hbm_gb_per_s_from_counters": round((2 * fk + wk) * 1024 / (tot / n), 1)}
    json.dump(out, open(f"profiles/{tag}_pix2pix_hbm_rates.json", "w"), indent=1)
    # what bench.py's pix2pix.roofline.traffic reads
    json.dump(out, open(f"profiles/{tag}_pix2pix_hbm_traffic.json", "w"), indent=1)
    json.dump(out, open("profiles/pix2pix_hbm_traffic_current.json", "w"), indent=1)
    for k, v in out.items():
        print(f"pix2pix {k:52s} {v['hbm_bytes_per_launch_corrected']/1e6:9.1f} MB/launch {v['hbm_gb_per_s_from_counters']:8.1f} GB/s")
