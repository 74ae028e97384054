"""Summarise rocprofv3 --pmc SQ / GRBM passes (scripts/profile_round.sh: <out>/pmc_sq1, pmc_sq2 and, for the pix2pix
step, p2p_pmc_sq1) into profiles/<tag>_sq_counters.json: per kernel group the per-launch sums, the wave-cycle split
(issuing / issue-stalled / parked) and the matrix-pipe utilisation:

    mfma_util_at_clock = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)
    mfma_flops         = SQ_INSTS_VALU_MFMA_MOPS_BF16 * 512

SQ_VALU_MFMA_BUSY_CYCLES is the sum over the SIMDs of the cycles their matrix pipe was held: it equals SQ_INSTS_MFMA x 16
for v_mfma_f32_16x16x32_bf16 and x 32 for 32x32x16 (checked here), i.e. MOPS / 2.  Round 2 read it as "saturated" because
the forward / data-gradient kernel and the weight-gradient kernel showed the SAME value per launch: they do -- both groups
average the same 104.6 GFLOP per launch over the same 17 layers.  GRBM_GUI_ACTIVE is reported summed over the 8 XCDs; the
quotient by the launch duration is the clock the chip held (it reads high on launches well under 0.3 ms)."""
import csv, json, os, sys
from collections import defaultdict

tag = sys.argv[1]
src = f"gpurun_out/prof_{tag}"
GROUPS = {"cfm": {"conv3x3 fwd + dgrad (conv3x3_dma16_kernel + conv3x3_stage_kernel + conv3x3_pers16_kernel)": ("conv3x3_dma16_kernel", "conv3x3_pers16_kernel", "conv3x3_stage_kernel"),
                  # per instantiation: the 128-wide tile of the wide layers, the 64-wide one-tile and
                  # persistent forms of the 64- / 128-channel layers at 256^2 / 128^2
                  "conv3x3_dma16_kernel<8, 32, 128, 2, 2, 4> (wide layers)": ("conv3x3_dma16_kernel<8, 32, 128, 2, 2, 4>",),
                  "conv3x3_dma16_kernel<8, 32, 64, 4, 1, 4>": ("conv3x3_dma16_kernel<8, 32, 64, 4, 1, 4>",),
                  "conv3x3_pers16_kernel (64-wide tiles, persistent walk)": ("conv3x3_pers16_kernel",),
                  "conv3x3_stage_kernel (256^2 level: filter resident / streaming, one barrier per chunk)": ("conv3x3_stage_kernel",),
                  "conv3x3_wgrad_dma_kernel": ("conv3x3_wgrad_dma_kernel",)},
          "p2p": {"convkxk_dma16_kernel (4x4 forward / data gradient / transposed)": ("convkxk_dma16_kernel",),
                  "convsm_kernel (inner levels: conv + norm / data gradient + norm backward in one launch)": ("convsm_kernel",),
                  "convflat_dma16_kernel (inner levels)": ("convflat_dma16_kernel",),
                  "conv2x2_wgrad_dma_kernel": ("conv2x2_wgrad_dma_kernel",)}}


def collect(dirs):
    acc = defaultdict(lambda: defaultdict(list))          # kernel -> counter -> values per dispatch
    dur = defaultdict(list)
    for d in dirs:
        path = None
        for root, _, files in os.walk(os.path.join(src, d)):
            for f in files:
                if f.endswith("counter_collection.csv"):
                    path = os.path.join(root, f)
        if not path:
            continue
        per = defaultdict(dict)
        for row in csv.DictReader(open(path)):
            per[(row["Dispatch_Id"], row["Kernel_Name"])][row["Counter_Name"]] = float(row["Counter_Value"])
            if "Start_Timestamp" in row and row.get("End_Timestamp"):
                per[(row["Dispatch_Id"], row["Kernel_Name"])]["_dur_ns"] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
        for (_, k), cs in per.items():
            for c, v in cs.items():
                acc[k][c].append(v)
    return acc


out = {"note": " ".join(__doc__.split())}
for which, dirs in (("cfm", ["pmc_sq1", "pmc_sq2"]), ("p2p", ["p2p_pmc_sq1"])):
    acc = collect(dirs)
    ndirs_with_dur = sum(1 for d in dirs if os.path.isdir(os.path.join(src, d)))
    for label, needles in GROUPS[which].items():
        ks = [k for k in acc if any(n in k for n in needles)]
        if not ks:
            continue
        tot = defaultdict(float)
        n = 0
        for k in ks:
            for c, vs in acc[k].items():
                tot[c] += sum(vs)
            n = max(n, sum(len(acc[k].get("SQ_WAVE_CYCLES", acc[k].get("SQ_INSTS_MFMA", []))) for k in [k]) + n)
        n = sum(len(next(iter(acc[k].values()))) for k in ks)
        e = {"launches": n}
        for c, v in sorted(tot.items()):
            if not c.startswith("_"):
                e[c + "_per_launch"] = round(v / n, 1)
        wc = tot.get("SQ_WAVE_CYCLES", 0.0)
        if wc:
            e["wave_cycles_split"] = {"active_inst": round(tot["SQ_ACTIVE_INST_ANY"] / wc, 3),
                                      "wait_inst": round(tot["SQ_WAIT_INST_ANY"] / wc, 3),
                                      "wait_any": round(tot["SQ_WAIT_ANY"] / wc, 3)}
        if tot.get("SQ_VALU_MFMA_BUSY_CYCLES") and tot.get("GRBM_GUI_ACTIVE"):
            cyc = tot["GRBM_GUI_ACTIVE"] / 8.0
            e["mfma_util_at_clock"] = round(tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0), 4)
            e["mfma_busy_cycles_per_instruction"] = round(tot["SQ_VALU_MFMA_BUSY_CYCLES"] / tot["SQ_INSTS_MFMA"], 2)
            if tot.get("_dur_ns"):
                dur = tot["_dur_ns"] / ndirs_with_dur
                e["avg_launch_us_under_pmc"] = round(dur / n / 1e3, 2)
                e["clock_ghz_from_grbm"] = round(cyc / dur, 3)
                e["mfma_tflops_from_counters"] = round(tot.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) * 512 / dur / 1e3, 1)
                e["mfma_frac_of_2.5PF"] = round(e["mfma_tflops_from_counters"] / 2500.0, 4)
        if wc and tot.get("SQ_WAIT_INST_LDS") is not None and "SQ_WAIT_INST_LDS" in tot:
            # second pass (pmc_sq2): what the issue stalls are made of, as fractions of the FIRST pass's wave cycles
            e["wait_inst_lds_frac_of_wave_cycles"] = round(tot["SQ_WAIT_INST_LDS"] / wc, 3)
        if tot.get("SQ_INSTS_MFMA"):
            for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS"):
                if tot.get(c):
                    e[c.lower() + "_per_mfma"] = round(tot[c] / tot["SQ_INSTS_MFMA"], 2)
        if tot.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_bank_conflict_frac_of_lds_cycles"] = round(tot.get("SQ_LDS_BANK_CONFLICT", 0.0) / tot["SQ_LDS_IDX_ACTIVE"], 4)
        out[label] = e
json.dump(out, open(f"profiles/{tag}_sq_counters.json", "w"), indent=1)
for k, v in out.items():
    if isinstance(v, dict):
        print(k, {x: v[x] for x in ("launches", "mfma_util_at_clock", "mfma_tflops_from_counters", "clock_ghz_from_grbm", "wave_cycles_split") if x in v})
