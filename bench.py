#!/usr/bin/env python3
"""Headline benchmark: paired 256x256 stain tiles/s for one full optimisation step of the
flow-matching U-Net (sample -> forward -> loss -> backward -> gradient all-reduce -> Adam).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1] as mapped by SURVEY.md section 8(d); the reference has no pix2pix
G+D, its stain models are all conditional flow matching): production U-Net
features [64,128,256,512,1024], 3x256x256 tiles, batch 16 per GPU, bf16 MFMA compute with fp32
accumulation / BatchNorm statistics / master weights, synthetic tiles U(-1,1), random-init weights.
Weak scaling: every rank steps its own 16 tiles; the only collective is the gradient all-reduce.

One JSON line on rank 0; see README / DESIGN.md for the roofline and cpu_baseline objects.
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FEATURES = [64, 128, 256, 512, 1024]
TILE = 256
BATCH_PER_GPU = 16
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes(batch: int, tile: int) -> dict:
    """Algorithmic HBM bytes per step of the bandwidth-bound passes (inputs + outputs at their storage
    width, bf16 activations / fp32 optimiser state) for the production U-Net, keyed like the --breakdown table."""
    f = FEATURES
    L = len(f)
    # BatchNorm'd conv outputs: two per encoder level, two per decoder level
    enc = sum(2 * f[l] * (tile >> l) ** 2 for l in range(L))
    dec = sum(2 * f[l] * (tile >> l) ** 2 for l in range(L - 1))
    bn_elems = batch * (enc + dec)
    pooled = batch * sum(f[l] * (tile >> (l + 1)) ** 2 for l in range(L - 1))
    up_out = batch * sum(f[l + 1] * (tile >> l) ** 2 for l in range(L - 1))          # up-sampled tensors (channels of the level below)
    params = 31_785_603
    return {"bn_relu_apply": 2 * (2 * bn_elems + pooled), "bn_relu_bwd": 2 * (5 * bn_elems + 2 * pooled),
            "upsample2x_fwd": 2 * (up_out + up_out // 4), "upsample2x_bwd": 2 * (up_out + up_out // 4),
            "adam_step_": 28 * params, "pack_conv3x3": (4 + 2 + 2) * params,
            "head_loss_fused": batch * tile * tile * (2 * 2 * f[0] + 2 * 4 * 3),
            "cfm_sample": batch * 3 * tile * tile * 4 * 4}


def cpu_baseline(seconds_budget: float = 30.0):
    """The CPU oracle (a port of the reference's torch path) timed on this box's host cores."""
    from oracle import unet_oracle as O
    from stain2stain_amd import FlowUNet
    torch.manual_seed(1984)
    P = {k: v.detach().clone() for k, v in FlowUNet(3, FEATURES, 3, 256).state_dict().items()}
    g = torch.Generator().manual_seed(1984)
    b = 2
    x0 = torch.rand(b, 3, TILE, TILE, generator=g) * 2 - 1
    x1 = torch.rand(b, 3, TILE, TILE, generator=g) * 2 - 1
    t = torch.rand(b, generator=g)
    # a 1-GPU box owns a 16-core share of the host; more threads than that only oversubscribe it
    threads = max(1, min(16, os.cpu_count() or 1))
    torch.set_num_threads(threads)
    O.train_steps(P, [(x0, x1, t)])          # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        O.train_steps(P, [(x0, x1, t)])
        n += 1
        el = time.perf_counter() - t0
        if n >= 6 or el > seconds_budget:       # mean of >= 5 timed steps at the production size
            break
    return {"value": round(b * n / el, 4), "unit": "tiles/s", "cores": threads, "kind": "port",
            "sample": f"same U-Net and step (fwd+bwd+Adam, fp32) at batch {b}, 1 warm-up + {n} timed steps, "
                      f"torch {torch.__version__} CPU, {threads} threads"}


def pix2pix_cpu_baseline(seconds_budget: float = 20.0):
    """The pix2pix G + D step of the torch-layer oracle (oracle/pix2pix_oracle.py: the checker of row a13, builder-authored
    because the reference has no such model) timed on this box's host cores: headline networks, batch 2, fp32."""
    from oracle import pix2pix_oracle as PO
    torch.manual_seed(1984)
    G, D = PO.OracleGenerator(), PO.OracleDiscriminator()
    og = torch.optim.Adam(G.parameters(), lr=2e-4, betas=(0.5, 0.999))
    od = torch.optim.Adam(D.parameters(), lr=2e-4, betas=(0.5, 0.999))
    g = torch.Generator().manual_seed(1984)
    b = 2
    src = torch.rand(b, 3, TILE, TILE, generator=g) * 2 - 1
    tgt = torch.rand(b, 3, TILE, TILE, generator=g) * 2 - 1
    threads = max(1, min(16, os.cpu_count() or 1))
    torch.set_num_threads(threads)
    PO.pix2pix_step(G, D, og, od, src, tgt)           # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        PO.pix2pix_step(G, D, og, od, src, tgt)
        n += 1
        el = time.perf_counter() - t0
        if n >= 6 or el > seconds_budget:
            break
    return {"value": round(b * n / el, 4), "unit": "tiles/s", "cores": threads, "kind": "port",
            "sample": f"same networks and G + D step (torch layers, fp32, two torch.optim.Adam) at batch {b}, 1 warm-up + "
                      f"{n} timed steps, torch {torch.__version__} CPU, {threads} threads"}


def pix2pix_flops(batch: int, tile: int, ngf: int = 64, ndf: int = 64, num_downs: int = 8) -> dict:
    """Algorithmic FLOP of one pix2pix G + D step (row a13), real channel counts (3-channel images, 1 logit channel):
    2 MAC per weight tap and output pixel; forward, data gradient and weight gradient of every layer that needs them."""
    ch = [ngf * min(2 ** i, 8) for i in range(num_downs)]
    g_fwd = 0.0
    for i in range(num_downs):                          # encoder: 4x4 stride-2 convolutions
        cin = 3 if i == 0 else ch[i - 1]
        g_fwd += 2.0 * (tile >> (i + 1)) ** 2 * ch[i] * 16 * cin
    for i in range(num_downs - 1, -1, -1):              # decoder: transposed convolutions (4 taps per output pixel)
        cin = ch[i] if i == num_downs - 1 else 2 * ch[i]
        cout = 3 if i == 0 else ch[i - 1]
        g_fwd += 2.0 * (tile >> i) ** 2 * cout * 4 * cin
    d_layers = [(6, ndf, tile // 2), (ndf, 2 * ndf, tile // 4), (2 * ndf, 4 * ndf, tile // 8),
                (4 * ndf, 8 * ndf, tile // 8 - 1), (8 * ndf, 1, tile // 8 - 2)]
    d_fwd = sum(2.0 * o * o * co * 16 * ci for ci, co, o in d_layers)
    # G: fwd + dgrad + wgrad.  D: forward on 2B (update) + B (generator pass); weight gradients on 2B; data gradients
    # on 2B (update, all but the first layer) + B (generator pass)
    g_first = 2.0 * (tile >> 1) ** 2 * ch[0] * 16 * 3                   # downs.0: its data gradient is never formed
    d_first = 2.0 * d_layers[0][2] ** 2 * d_layers[0][1] * 16 * d_layers[0][0]
    # what the engine LAUNCHES in the forward / data-gradient group, real channel counts: G forward + data gradients (all
    # but downs.0); D forward on 2B + B; D data gradients on 2B (all but c1: the update needs no input gradient) + B (all)
    conv_fd = batch * ((2 * g_fwd - g_first) + 3 * d_fwd + 2 * (d_fwd - d_first) + d_fwd)
    return {"generator": batch * 3 * g_fwd, "discriminator": batch * (3 * d_fwd + 2 * d_fwd + 3 * d_fwd),
            "g_fwd_per_tile": g_fwd, "d_fwd_per_tile": d_fwd, "conv_fwd_dgrad_launched": conv_fd,
            "generator_launched": batch * (3 * g_fwd - g_first)}


def _committed_traffic(fname: str, prefix: str):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary (FETCH_SIZE / WRITE_SIZE in
    separate passes of this same command, gfx950 correction applied: scripts/summarise_profiles.py), or None."""
    path = os.path.join(ROOT, "profiles", fname)
    try:
        tj = json.load(open(path))
        return next(v["hbm_bytes_per_launch_corrected"] for k, v in tj.items() if k.startswith(prefix))
    except Exception:  # noqa: BLE001
        return None


def rccl_object(dev, world: int, use_dist: bool):
    """Which collective backend the line ran on and which ranks took part (an all-gather of the rank ids over the very
    process group the gradient exchange uses), so that a scaling record proves its N ranks."""
    if not use_dist:
        return {"backend": None, "world": 1, "ranks_seen": [0]}
    mine = torch.tensor([dist.get_rank()], device=dev, dtype=torch.int64)
    seen = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(seen, mine)
    backend = dist.get_backend()
    return {"backend": "rccl (torch.distributed 'nccl')" if backend == "nccl" else backend, "world": dist.get_world_size(),
            "ranks_seen": sorted(int(t) for t in seen)}


def _launch_note(graph: bool) -> str:
    return ("one hipGraph replay per optimisation step (captured on the second step; event-bracketed steps run eagerly)"
            if graph else "eager: every kernel launched from Python (weight gradients on a side HIP stream)")


def _other_leg(trainer, step_fn, steps: int, use_dist: bool):
    """ms per step of the SAME trainer in the OTHER launch mode (graph replay <-> eager launches): a short second timed
    loop under the same contract (barrier + synchronize on both sides).  Both modes run the same kernels to the same
    bits (tests/test_graph_step_gpu.py); the headline loop uses the faster one on this stack, the line reports both."""
    n = max(5, steps // 2)
    was = trainer.graph
    trainer.graph, trainer.overlap_wgrad = not was, True
    pause = _GcPause()
    step_fn(0)
    step_fn(1)                    # (the second step of a shape captures)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        step_fn(i + 2)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / n
    pause.resume()
    trainer.close()               # the captured graph goes now (owned lifetime), not at interpreter exit
    trainer.graph = was
    return round(ms, 3)


def sample_leg(dev) -> dict:
    """BASELINE.json configs[3] inside the default line: the eval-mode network (BatchNorm folded into the conv epilogue),
    50 fixed Euler steps, batch 32 -- and batch 1, the reference's own use (src/infer_simple_flowmatching.py:73-83) --
    launched eagerly and as one replayed hipGraph per Euler step.  Synchronised wall time of whole solves."""
    from stain2stain_amd import FlowUNet, euler_generate
    torch.manual_seed(1984)
    net = FlowUNet(3, FEATURES, 3, 256).to(dev).eval()
    g = torch.Generator().manual_seed(1984)
    out = {"metric": f"{TILE}x{TILE} tiles/sec sampled (50 Euler steps, eval-mode network)", "unit": "tiles/s", "dtype": "bf16",
           "config": {"workload": f"CFM U-Net {FEATURES} 3x{TILE}x{TILE}, 50 Euler steps t_k = k/50 (BASELINE.json configs[3]: "
                                  "batch 32; batch 1 = the reference's own use)"}}
    for B in (32, 1):
        src = (torch.rand(B, 3, TILE, TILE, generator=g) * 2 - 1).to(dev)
        res = {}
        for graph in (False, True):
            x = euler_generate(net, src, 50, graph=graph)            # warm-up / capture
            torch.cuda.synchronize()
            n = 2 if B == 32 else 5
            t0 = time.perf_counter()
            for _ in range(n):
                x = euler_generate(net, src, 50, graph=graph)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3 / n
            res["graph" if graph else "eager"] = {"ms_per_solve": round(ms, 3), "ms_per_euler_step": round(ms / 50, 4),
                                                  "tiles_per_s": round(B * 1e3 / ms, 2), "solves_timed": n,
                                                  "finite": bool(torch.isfinite(x).all())}
        out[f"batch{B}"] = res
        cap = getattr(net, "_s2s_euler_graph", None)
        if cap is not None:
            cap.close()
    out["value"] = out["batch32"]["graph"]["tiles_per_s"]
    return out


def fp32_parity_leg(dev, B: int) -> dict:
    """The same optimisation step in the fp32 parity mode (three-way bf16 split on the MFMA path, fp32 storage): the
    mode the 1e-3 parity tests against the CPU oracle run in (tests/test_e2e_gpu.py), timed beside the bf16 headline."""
    from stain2stain_amd import CFMTrainer, FlowUNet
    torch.manual_seed(1984)
    net = FlowUNet(3, FEATURES, 3, 256, precision="fp32").to(dev).train()
    tr = CFMTrainer(net, lr=1e-4, weight_decay=1e-5)
    g = torch.Generator().manual_seed(1984)
    x0 = (torch.rand(B, 3, TILE, TILE, generator=g) * 2 - 1).to(dev)
    x1 = (torch.rand(B, 3, TILE, TILE, generator=g) * 2 - 1).to(dev)
    t = torch.rand(B, generator=g).to(dev)
    tr.step(x0, x1, t)
    torch.cuda.synchronize()
    n = 4
    t0 = time.perf_counter()
    for _ in range(n):
        loss = tr.step(x0, x1, t)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / n
    return {"metric": f"paired {TILE}x{TILE} stain tiles/sec (full CFM optimisation step, fp32 parity mode)",
            "value": round(B * 1e3 / ms, 2), "unit": "tiles/s", "ms_per_step": round(ms, 3), "steps": n,
            "dtype": "fp32 (3-way bf16 split on MFMA, fp32 storage and accumulation)", "final_loss": round(float(loss), 6),
            "note": "parity with the reference's fp32 CPU path <= 1e-3 is stated (and tested) for THIS mode; the bf16 "
                    "headline is held to 4e-3 per op against torch fp32 on bf16-rounded operands"}


def multitask_flops(tile: int) -> float:
    """Algorithmic FLOP per tile of the multitask step: the shared encoder twice, the flow decoder and the mask decoder
    (same Up blocks, a 1-channel head), forward + data gradient + weight gradient = 3 x forward."""
    f = FEATURES
    enc = 2.0 * 9 * (3 * f[0] + f[0] * f[0]) * tile * tile
    for l in range(1, len(f)):
        enc += 2.0 * 9 * (f[l - 1] * f[l] + f[l] * f[l]) * (tile >> l) ** 2
    dec = 0.0
    for l in range(len(f) - 2, -1, -1):
        dec += 2.0 * 9 * ((f[l + 1] + f[l]) * f[l] + f[l] * f[l]) * (tile >> l) ** 2
    heads = 2.0 * f[0] * (3 + 1) * tile * tile
    return 3.0 * (2 * enc + 2 * dec + heads)


def multitask_bench(args, dev, rank: int, world: int, use_dist: bool, tile: int, batch: int, steps: int, warmup: int) -> dict:
    """BASELINE.json configs[4] (row f2): the multitask step -- shared encoder on xt and on the source, flow head, mask
    head, Dice + BCE, one Adam over encoder + flow_decoder + seg_decoder (conditional_flow_matching_multitask.py:204-257,
    391-417) -- as the fused ``MultiTaskTrainer``; 512 x 512 tiles, batch 8 per GPU.  Same timing contract as the headline."""
    from stain2stain_amd import FlowMatchingDecoder, MultiTaskTrainer, SegmentationDecoder, SharedEncoder, ops
    torch.manual_seed(1984)
    enc = SharedEncoder(3, FEATURES, precision=args.precision).to(dev).train()
    fdec = FlowMatchingDecoder(FEATURES[-1], list(FEATURES[:-1][::-1]), 3, 256, precision=args.precision).to(dev).train()
    sdec = SegmentationDecoder(FEATURES[-1], list(FEATURES[:-1][::-1]), 1, precision=args.precision).to(dev).train()
    tr = MultiTaskTrainer(enc, fdec, sdec, time_emb_dim=256, lr=1e-4, weight_decay=1e-5,
                          sharded_optimizer=args.sharded_optimizer)
    g = torch.Generator().manual_seed(1984 + rank)
    pool = [((torch.rand(batch, 3, tile, tile, generator=g) * 2 - 1).to(dev), (torch.rand(batch, 3, tile, tile, generator=g) * 2 - 1).to(dev),
             (torch.rand(batch, 1, tile, tile, generator=g) > 0.7).float().to(dev)) for _ in range(2)]
    ts = [torch.rand(batch, generator=g).to(dev) for _ in range(warmup + steps)]
    for i in range(warmup):
        tr.step(*pool[i % 2], ts[i])
    pause = _GcPause()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    every = max(1, args.event_every)
    prof, timed_steps, loss = [], 0, None
    t0 = time.perf_counter()
    for i in range(steps):
        sampled = i % every == min(every // 2, steps - 1)
        tr.overlap_wgrad = not sampled
        if sampled:
            ops.profile_start(("conv3x3_mfma", "conv3x3_wgrad_mfma"))
        loss = tr.step(*pool[(warmup + i) % 2], ts[warmup + i])
        if sampled:
            prof += ops.profile_stop()
            timed_steps += 1
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    pause.resume()
    if use_dist:
        el = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el)
    agg = {}
    for name, work, e0, e1 in prof:
        a = agg.setdefault(name, [0, 0.0, 0.0])
        a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e-3; a[2] += work
    n_l, t_l, f_l = agg.get("conv3x3_mfma", [0, 1e-9, 0.0])
    fl = multitask_flops(tile)
    return {
        "metric": f"paired {tile}x{tile} stain tiles/sec (multitask optimisation step: flow + mask heads on a shared encoder)",
        "value": round(batch * world * steps / elapsed, 3), "unit": "tiles/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": round(elapsed * 1e3 / steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
        "config": {"workload": f"multitask U-Net {FEATURES}: shared encoder x 2 + flow decoder + mask decoder, 3x{tile}x{tile} "
                               f"tiles, batch {batch}/GPU, CFM MSE + 1.0 x (0.5 Dice + 0.5 BCE), one fused Adam(1e-4, wd 1e-5) "
                               "(BASELINE.json configs[4]; the reference's 4-domain any2any network is third-party, "
                               "SURVEY 8d: the in-repo multitask model stands in)",
                   "global_batch": batch * world, "tile": tile, "parallelism": f"dp{world}",
                   "final_loss": round(float(loss), 6),
                   "grad_exchange": (tr.fp.bucketer.mode if tr.fp.bucketer.enabled else "none"),
                   "algorithmic_gflop_per_tile": round(fl / 1e9, 1),
                   "step_tflops": round(fl * batch * steps / elapsed / 1e12, 1)},
        "roofline": {"bound": "mfma", "kernel": "conv3x3 fwd + dgrad launches (conv3x3_dma16_kernel; conv3x3_stage_kernel on the 256^2 level)",
                     "achieved": round(f_l / t_l / 1e12, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(f_l / t_l / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None,
                     "launches_per_step": n_l // max(timed_steps, 1), "launches_timed": n_l,
                     "avg_launch_ms": round(t_l * 1e3 / max(n_l, 1), 4)},
        "kernels": {k: {"launches": v[0], "ms_per_step": round(v[1] * 1e3 / max(timed_steps, 1), 4),
                        **({"tflops": round(v[2] / v[1] / 1e12, 1)} if v[2] else {})}
                    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}}


def dropin_leg(dev, B: int, steps: int, fused_ms: float) -> dict:
    """The path a user of the reference gets after the ``_target_`` swap: ``FlowUNet`` under autograd inside
    ``ConditionalFlowMatchingModule`` driven the way Lightning's automatic optimisation drives the reference's
    LightningModule (src/models/conditional_flow_matching.py:76-88 ``training_step``, :112-131 ``configure_optimizers``):
    ``optimizer.zero_grad(); loss = module.training_step(batch, i); loss.backward(); optimizer.step()`` -- same network,
    batch, precision and timing contract as the headline, which runs the fused trainer instead.  Timed with the optimiser
    the reference's config names (``torch.optim.Adam(lr=1e-4, weight_decay=1e-5)``) and with ``stain2stain_amd.FusedAdam``
    (the same update as one HIP launch, selected by the same ``_target_`` mechanism)."""
    from functools import partial
    from stain2stain_amd import ConditionalFlowMatchingModule, FlowUNet, FusedAdam
    out = {"metric": f"paired {TILE}x{TILE} stain tiles/sec (drop-in modules under autograd: zero_grad, training_step, "
                     "backward, optimizer.step)", "unit": "tiles/s", "dtype": "bf16", "steps": steps}
    g = torch.Generator().manual_seed(1984)
    pool = [((torch.rand(B, 3, TILE, TILE, generator=g) * 2 - 1).to(dev), (torch.rand(B, 3, TILE, TILE, generator=g) * 2 - 1).to(dev))
            for _ in range(4)]
    for name, opt_cls in (("fused_adam", FusedAdam), ("torch_adam", torch.optim.Adam)):
        torch.manual_seed(1984)
        net = FlowUNet(3, FEATURES, 3, 256).to(dev).train()
        mod = ConditionalFlowMatchingModule(net, optimizer=partial(opt_cls, lr=1e-4, weight_decay=1e-5))
        opt = mod.configure_optimizers()["optimizer"]

        def one(i):
            opt.zero_grad()
            loss = mod.training_step(pool[i % 4], i)
            loss.backward()
            opt.step()
            return loss

        for i in range(3):
            one(i)
        pause = _GcPause()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            loss = one(3 + i)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / steps
        pause.resume()
        out[name] = {"ms_per_step": round(ms, 3), "tiles_per_s": round(B * 1e3 / ms, 2), "final_loss": round(float(loss), 6)}
        del net, mod, opt
        torch.cuda.empty_cache()
    out["value"] = out["fused_adam"]["tiles_per_s"]
    out["ms_per_step"] = out["fused_adam"]["ms_per_step"]
    out["vs_fused_trainer"] = round(out["ms_per_step"] / fused_ms, 4)
    out["config"] = {"workload": f"CFM U-Net {FEATURES} 3x{TILE}x{TILE}, batch {B}, stain2stain_amd.FlowUNet inside "
                                 "ConditionalFlowMatchingModule; optimizer = stain2stain_amd.FusedAdam(lr 1e-4, wd 1e-5) "
                                 "(torch_adam: torch.optim.Adam, the class the reference's config names)"}
    return out


class _GcPause:
    """No cyclic garbage collection inside a timed region.  A full (generation 2) collection walks every object torch
    has created -- 60-80 ms here -- and when it fires is a matter of allocation counts: with the package loaded from its
    bytecode cache it fell into the 20 timed steps of the driver's run (+3.2 ms per step: 1440 instead of 2030 tiles/s),
    with the package compiled from source it fell into the warm-up.  Reference counting frees the step's tensors either
    way; training loops that care do the same (gc.freeze() after set-up)."""

    def __init__(self):
        self.was = gc.isenabled() and os.environ.get("S2S_BENCH_GC") != "1"
        if self.was:
            gc.collect()
            gc.disable()

    def resume(self) -> None:
        if self.was:
            gc.enable()


def pix2pix_bench(args, dev, rank: int, world: int, use_dist: bool) -> dict:
    """BASELINE.json configs[1] as literally worded: 8-level U-Net generator + 70x70 PatchGAN discriminator, batch 16 per
    GPU, bf16, GAN(BCE) + 100 L1, two Adam(2e-4, 0.5/0.999) -- the fused HIP engine (stain2stain_amd.Pix2PixTrainer),
    no torch compute in the step.  Row a13: not in the reference, parity against the torch-layer oracle only.
    Same timing contract as the main line: barrier + synchronize on both sides, max over ranks."""
    from stain2stain_amd import PatchGANDiscriminator, Pix2PixGenerator, Pix2PixTrainer, ops
    torch.manual_seed(1984)
    G, D = Pix2PixGenerator().to(dev), PatchGANDiscriminator().to(dev)
    tr = Pix2PixTrainer(G, D, lr=2e-4, betas=(0.5, 0.999), lambda_l1=100.0, precision=args.precision,
                        sharded_optimizer=args.sharded_optimizer)
    graph_ok = not (use_dist and dist.get_backend() != "nccl")       # only RCCL collectives can be captured
    tr.graph = args.train_graph and graph_ok
    B = args.batch
    g = torch.Generator().manual_seed(1984 + rank)
    # four distinct synthetic batches rotate through the loop (a fixed batch would let the activations sparsify)
    data = [((torch.rand(B, 3, TILE, TILE, generator=g) * 2 - 1).to(dev), (torch.rand(B, 3, TILE, TILE, generator=g) * 2 - 1).to(dev))
            for _ in range(4)]
    steps, warmup = args.steps, args.warmup
    for i in range(warmup):
        tr.step(*data[i % 4])
    gc_pause = _GcPause()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    convs = ("convkxk_mfma", "convkxk_wgrad_mfma")
    every = 1 if args.breakdown else max(1, args.event_every)
    prof, timed_steps = [], 0
    t0 = time.perf_counter()
    losses = None
    for i in range(steps):
        sampled = i % every == min(every // 2, steps - 1)    # mid-cycle: not the first step after the warm-up barrier
        tr.overlap_wgrad = not sampled          # bracketed steps on one stream (see the note in the CFM loop)
        if sampled:
            ops.profile_start(None if args.breakdown else convs)
        losses = tr.step(*data[(warmup + i) % 4])
        if sampled:
            prof += ops.profile_stop()
            timed_steps += 1
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gc_pause.resume()
    if use_dist:
        el = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el)
    ld, lg = tr.loss_values(losses)
    # (the second launch mode is timed at world size 1 only: a captured multi-rank RCCL exchange has never run on hardware
    #  available to the builder, and nothing untested goes between the timed region and the JSON line of a scaling run)
    other_ms = (_other_leg(tr, lambda i: tr.step(*data[i % 4]), steps, use_dist)
                if (graph_ok and world == 1 and not args.no_extras) else None)
    agg, by_net = {}, {}
    for name, work, e0, e1 in prof:
        base, _, net = name.partition("@")          # brackets are tagged with the network whose pass issued them
        dt_s = e0.elapsed_time(e1) * 1e-3
        a = agg.setdefault(base, [0, 0.0, 0.0])
        a[0] += 1; a[1] += dt_s; a[2] += work
        b = by_net.setdefault((base, net), [0, 0.0])
        b[0] += 1; b[1] += dt_s
    n_l, t_l, _ = agg.get("convkxk_mfma", [0, 1e-9, 0.0])
    traffic = _committed_traffic("pix2pix_hbm_traffic_current.json", "convkxk") if (
        B == BATCH_PER_GPU and args.precision == "bf16" and TILE == 256) else None
    fl = pix2pix_flops(B, TILE)
    step_flop = fl["generator"] + fl["discriminator"]
    ts = max(timed_steps, 1)
    f_l = fl["conv_fwd_dgrad_launched"] * ts       # REAL channel counts (3-channel images, 1 logit), not the padded 8
    # the generator's conv stack as north_star words its target: forward + data gradient + weight gradient (fold
    # included) launches of the 16 generator layers over their algorithmic FLOP
    g_t = sum(v[1] for (base, net), v in by_net.items() if net == "G" and base in ("convkxk_mfma", "convkxk_wgrad_mfma"))
    g_n = sum(v[0] for (base, net), v in by_net.items() if net == "G" and base in ("convkxk_mfma", "convkxk_wgrad_mfma"))
    gen_tflops = fl["generator_launched"] * ts / max(g_t, 1e-9) / 1e12
    return {
        "metric": f"paired {TILE}x{TILE} stain tiles/sec (pix2pix G+D optimisation step)",
        "value": round(B * world * steps / elapsed, 3), "unit": "tiles/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": round(elapsed * 1e3 / steps, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
        "config": {"workload": f"pix2pix 8-level U-Net G + 70x70 PatchGAN D, 3x{TILE}x{TILE}, batch {B}/GPU, GAN(BCE) + 100 L1, "
                               "two fused Adam(2e-4, 0.5/0.999), D then G update; every launch a HIP kernel "
                               "(row a13: not in the reference, parity vs the torch-layer oracle)",
                   "global_batch": B * world, "parallelism": f"dp{world}", "loss_d": round(ld, 5), "loss_g": round(lg, 4),
                   "loss_bits": [float(v).hex() for v in losses.detach().cpu().tolist()],
                   "grad_exchange": (tr.pG.bucketer.mode if tr.pG.bucketer.enabled else "none"),
                   "buckets_mb": [round((hi - lo) * 4 / 2 ** 20, 2) for _, lo, hi in tr.pG.bucketer.buckets],
                   "algorithmic_gflop_per_step": round(step_flop / 1e9, 1),
                   "step_tflops": round(step_flop * steps / elapsed / 1e12, 1),
                   "launch": _launch_note(tr.graph),
                   ("eager_ms_per_step" if tr.graph else "graph_ms_per_step"): other_ms},
        "roofline": {"bound": "mfma", "kernel": "convkxk_dma16_kernel (forward / data-gradient / transposed launches)",
                     "achieved": round(f_l / t_l / 1e12, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(f_l / t_l / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": traffic,
                     "traffic_unit": "HBM bytes per launch, average over the conv forward / data-gradient launches "
                                     "(rocprofv3 FETCH_SIZE*2 + WRITE_SIZE, profiles/pix2pix_hbm_traffic_current.json)",
                     "launches_per_step": n_l // max(timed_steps, 1), "launches_timed": n_l,
                     "avg_launch_ms": round(t_l * 1e3 / max(n_l, 1), 4),
                     "note": "FLOP = 2 MAC per tap and output pixel on the REAL channel counts (3-channel images, one "
                             "logit channel) of the launches the step issues; brackets on one stream"},
        "generator_conv_frac": round(gen_tflops / MFMA_BF16_PEAK_TFLOPS, 4),
        "generator_conv": {"tflops": round(gen_tflops, 1), "launches_per_step": g_n // ts,
                           "ms_per_step": round(g_t * 1e3 / ts, 4),
                           "gflop_per_step": round(fl["generator_launched"] / 1e9, 1),
                           "what": "forward + data-gradient + weight-gradient launches of the 16 generator layers "
                                   "(HIP-event brackets), the stack BASELINE.json's 40 % target names"},
        "launches_per_step": (sum(v[0] for v in agg.values()) // ts) if args.breakdown else None,
        "kernels": {k: {"launches": v[0], "ms_per_step": round(v[1] * 1e3 / max(timed_steps, 1), 4),
                        **({"tflops": round(v[2] / v[1] / 1e12, 1)} if v[2] else {})}
                    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="tiles per GPU (default: the headline 16)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--tile", type=int, default=TILE, help="tile edge (default 256; 512 = BASELINE.json configs[4])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pix2pix", action="store_true", help="skip the second timed loop (the pix2pix G + D step)")
    ap.add_argument("--sync-batchnorm", action="store_true",
                    help="BatchNorm statistics over the global batch (configs/trainer/ddp.yaml:9; off in the stain experiments)")
    ap.add_argument("--sharded-optimizer", action="store_true",
                    help="gradient reduce-scatter + Adam on 1/world of every bucket + parameter all-gather instead of "
                         "all-reduce + full Adam (same results; ddp.GradBucketer mode 'reduce_scatter')")
    ap.add_argument("--breakdown", action="store_true",
                    help="HIP events around EVERY op (per-kernel table; costs ~1 ms/step of host time)")
    ap.add_argument("--event-every", type=int, default=16,
                    help="bracket the conv launches of every N-th timed step with HIP events for the roofline leg "
                         "(1 = every step).  A bracketed step runs on ONE stream so that a bracket times its kernel alone: "
                         "it takes ~0.9 ms longer than an overlapped step, so N = 8 cost the line 1.4 %%, 16 costs 0.7 %%")
    ap.add_argument("--mode", default="train", choices=["train", "sample", "pix2pix", "multitask"],
                    help="train = the headline optimisation step (default); sample = BASELINE.json configs[3], "
                         "50 fixed Euler steps of the eval-mode network on a batch of 32 tiles (secondary line)")
    ap.add_argument("--euler-steps", type=int, default=50)
    ap.add_argument("--graph", action="store_true",
                    help="--mode sample: replay one hipGraph-captured Euler step instead of launching its ~60 kernels")
    ap.add_argument("--train-graph", action="store_true",
                    help="training modes: time the replay of one captured hipGraph per optimisation step in the headline "
                         "loop (default: eager launches, which overlap the side-stream weight gradients better on this "
                         "ROCm; the line reports the other mode's ms/step beside the headline either way)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary legs of the default line (\"sample\": configs[3], \"fp32_parity\")")
    ap.add_argument("--h2d", action="store_true",
                    help="PCIe-inclusive variant for DESIGN.md: every step copies a fresh uint8 batch from pinned host "
                         "memory and runs the GPU crop/flip/normalise kernel before the optimisation step")
    args = ap.parse_args()
    globals()["TILE"] = args.tile

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    local %= max(1, torch.cuda.device_count())      # more ranks than GPUs (rehearsals on a one-GPU box): share devices
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if os.environ.get("S2S_BENCH_OWN_STREAM") == "1":
        # everything on a non-blocking stream of our own instead of the legacy default stream (experiments with
        # S2S_WGRAD_CUS: a CU-masked stream is a blocking stream, which synchronises implicitly with the default one)
        torch.cuda.set_stream(torch.cuda.Stream(dev))
    use_dist = world > 1 or os.environ.get("S2S_FORCE_DDP") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = os.environ.get("S2S_BENCH_BACKEND", "nccl")       # "gloo": rehearse N ranks on fewer GPUs (RCCL wants
        if backend == "nccl":                                       # one device per rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from stain2stain_amd import CFMTrainer, FlowUNet, euler_generate, ops

    if args.mode == "multitask":
        # BASELINE.json configs[4] defaults: 512 x 512 tiles, batch 8 per GPU (--tile / --batch override them)
        tile = args.tile if args.tile != 256 else 512
        batch = args.batch if args.batch != BATCH_PER_GPU else 8
        res = multitask_bench(args, dev, rank, world, use_dist, tile, batch, args.steps, args.warmup)
        if rank == 0:
            _emit(json.dumps(res))
        if use_dist:
            dist.destroy_process_group()
        return
    if args.mode == "pix2pix":
        res = pix2pix_bench(args, dev, rank, world, use_dist)
        if rank == 0:
            _emit(json.dumps(res))
        if use_dist:
            dist.destroy_process_group()
        return

    torch.manual_seed(1984)
    net = FlowUNet(3, FEATURES, 3, 256, precision=args.precision).to(dev).train()
    if args.mode == "sample":
        # replicas only: every rank integrates its own batch, no collective on the data path
        B = 32 if args.batch == BATCH_PER_GPU else args.batch
        g = torch.Generator().manual_seed(1984 + rank)
        src = (torch.rand(B, 3, TILE, TILE, generator=g) * 2 - 1).to(dev)
        net.eval()
        for _ in range(max(1, args.warmup // 3)):
            euler_generate(net, src, args.euler_steps, graph=args.graph)
        gc_pause = _GcPause()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        prof, timed_steps = [], 0
        t0 = time.perf_counter()
        for i in range(args.steps):
            sampled = i % max(1, args.event_every) == min(max(1, args.event_every) // 2, args.steps - 1) and not args.graph     # (no event brackets inside a graph)
            if sampled:
                ops.profile_start(("conv3x3_mfma",))
            out = euler_generate(net, src, args.euler_steps, graph=args.graph)
            if sampled:
                prof += ops.profile_stop()
                timed_steps += 1
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        gc_pause.resume()
        if use_dist:
            el = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
            elapsed = float(el)
        if rank == 0:
            n_l = max(len(prof), 1)
            t_l = max(sum(e0.elapsed_time(e1) for _, _, e0, e1 in prof) * 1e-3, 1e-12)
            f_l = sum(w for _, w, _, _ in prof)
            timed_steps = max(timed_steps, 1)
            _emit(json.dumps({
                "metric": f"256x256 tiles/sec sampled ({args.euler_steps} Euler steps, eval-mode network)",
                "value": round(B * world * args.steps / elapsed, 3), "unit": "tiles/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed * 1e3 / args.steps, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision,
                "data": "synthetic",
                "config": {"workload": f"CFM U-Net {FEATURES} 3x256x256, batch {B}/GPU, {args.euler_steps} Euler "
                                       "steps (BatchNorm folded into the conv epilogue)"
                                       + (", one hipGraph replay per step" if args.graph else ", eager launches"),
                           "ms_per_euler_step": round(elapsed * 1e3 / args.steps / args.euler_steps, 4),
                           "global_batch": B * world, "parallelism": f"replicas x{world}",
                           "finite": bool(torch.isfinite(out).all())},
                "roofline": {"bound": "mfma", "kernel": "conv3x3_dma16_kernel (forward launches)",
                             "achieved": round(f_l / t_l / 1e12, 2), "peak": MFMA_BF16_PEAK_TFLOPS,
                             "unit": "TFLOP/s", "frac": round(f_l / t_l / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                             "traffic": None, "launches_per_step": n_l // timed_steps,
                             "launches_timed": n_l, "avg_launch_ms": round(t_l * 1e3 / n_l, 4)}}))
        if use_dist:
            dist.destroy_process_group()
        return
    graph_ok = (not args.h2d and not args.breakdown and not args.sync_batchnorm
                and not (use_dist and dist.get_backend() != "nccl"))
    use_graph = args.train_graph and graph_ok
    trainer = CFMTrainer(net, lr=1e-4, weight_decay=1e-5, sharded_optimizer=args.sharded_optimizer,
                         sync_batchnorm=args.sync_batchnorm, graph=use_graph)
    g = torch.Generator().manual_seed(1984 + rank)
    B = args.batch
    # four distinct synthetic batches rotate through the loop (on one fixed batch the activations sparsify as training
    # proceeds and the clock-bound conv kernels speed up, which would flatter a long run)
    pool = [((torch.rand(B, 3, TILE, TILE, generator=g) * 2 - 1).to(dev), (torch.rand(B, 3, TILE, TILE, generator=g) * 2 - 1).to(dev))
            for _ in range(4)]
    x0, x1 = pool[0]
    ts = [torch.rand(B, generator=g).to(dev) for _ in range(args.warmup + args.steps)]

    feed = None
    if args.h2d:
        from stain2stain_amd import data as s2s_data
        import random
        SRC = TILE + 32                                    # decoded images a little larger than the crop
        host = [torch.randint(0, 256, (B, SRC, SRC, 3), dtype=torch.uint8, generator=g).pin_memory() for _ in range(4)]
        rng = random.Random(1984 + rank)

        side = torch.cuda.Stream(device=dev)              # copies + preparation of batch i+1 overlap step i
        pending = {}

        def stage(i):
            with torch.cuda.stream(side):
                hs, ht = host[(2 * i) % 4], host[(2 * i + 1) % 4]
                ds, dtg = hs.to(dev, non_blocking=True), ht.to(dev, non_blocking=True)
                prm = s2s_data.sample_crop_flip_params(B, (SRC, SRC), TILE, rng)
                a, b = s2s_data.paired_crop_flip_normalize(ds, dtg, prm, TILE)
                ev = torch.cuda.Event()
                ev.record(side)
            pending[i] = (a, b, ev)

        def feed(i):
            if i not in pending:
                stage(i)
            a, b, ev = pending.pop(i)
            torch.cuda.current_stream().wait_event(ev)
            a.record_stream(torch.cuda.current_stream()); b.record_stream(torch.cuda.current_stream())
            stage(i + 1)
            return a, b

    for i in range(args.warmup):
        x0, x1 = feed(i) if feed else pool[i % 4]
        trainer.step(x0, x1, ts[i])
    gc_pause = _GcPause()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    # live HIP-event timing of the dominant kernels inside the timed region.  An event pair is two marker packets on
    # the launch stream (~5 us each side of the kernel, rocprofv3 kernel trace), i.e. ~0.3-0.5 ms per step if all
    # 51 conv launches of every step are bracketed; so the brackets go around every launch of every
    # --event-every'th step (default 16; 1 = every step) and the per-launch average is taken over those.
    every = 1 if args.breakdown else max(1, args.event_every)
    only = None if args.breakdown else ("conv3x3_mfma", "conv3x3_wgrad_mfma")
    prof, timed_steps = [], 0
    t0 = time.perf_counter()
    loss = None
    ms0 = dict(torch.cuda.memory_stats(dev)) if os.environ.get("S2S_BENCH_MEMSTAT") else None
    for i in range(args.steps):
        x0, x1 = feed(args.warmup + i) if feed else pool[(args.warmup + i) % 4]
        sampled = i % every == min(every // 2, args.steps - 1)    # mid-cycle: not the first step after the warm-up barrier
        # the bracketed steps run on ONE stream, so that an event pair times its kernel alone; all other steps overlap
        # the weight gradients with the bandwidth-bound backward passes on a side stream (engine.run_on_side)
        trainer.overlap_wgrad = not sampled
        if sampled:
            ops.profile_start(only)
        loss = trainer.step(x0, x1, ts[args.warmup + i])
        if sampled:
            prof += ops.profile_stop()
            timed_steps += 1
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    other_ms = None
    if graph_ok and world == 1 and not args.no_extras:      # (world size 1 only: see pix2pix_bench)
        other_ms = _other_leg(trainer, lambda i: trainer.step(*pool[i % 4], ts[i % len(ts)]), args.steps, use_dist)
    gc_pause.resume()
    if ms0 is not None:
        ms1 = torch.cuda.memory_stats(dev)
        keys = ("num_device_alloc", "num_device_free", "num_alloc_retries", "reserved_bytes.all.current",
                "reserved_bytes.all.peak", "allocated_bytes.all.peak", "segment.all.current", "allocation.all.allocated")
        print("memstat", {k: (ms0.get(k), ms1.get(k)) for k in keys}, file=sys.stderr)
    if use_dist:
        el = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el)

    if rank == 0:
        # per-kernel time from the HIP events recorded around every launch of the timed region
        agg = {}
        for name, work, e0, e1 in prof:
            a = agg.setdefault(name, [0, 0.0, 0.0])
            a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e-3; a[2] += work
        dom = "conv3x3_mfma"          # forward + data-gradient implicit-GEMM kernel
        # HBM bytes per launch of that kernel come from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate
        # runs of this same command, gfx950 correction applied); the committed summary is read back here
        traffic = None
        if args.batch == BATCH_PER_GPU and args.precision == "bf16" and TILE == 256:
            traffic = _committed_traffic("hbm_traffic_current.json", "conv3x3_mfma")
        n_l, t_l, f_l = agg[dom]
        achieved = f_l / t_l / 1e12
        ab = algorithmic_bytes(B, TILE)
        kernels = {k: {"launches": v[0], "ms_per_step": round(v[1] * 1e3 / timed_steps, 4),
                       **({"tflops": round(v[2] / v[1] / 1e12, 1)} if v[2] else {}),
                       **({"algorithmic_gb_per_s": round(ab[k] * timed_steps / v[1] / 1e9, 0)} if k in ab else {})}
                   for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}
        out = {
            "metric": f"paired {TILE}x{TILE} stain tiles/sec (full CFM optimisation step)",
            "value": round(B * world * args.steps / elapsed, 3),
            "unit": "tiles/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed * 1e3 / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic" + (" (uint8 tiles copied from pinned host memory every step)"
                                                           if args.h2d else ""),
            "config": {"workload": f"CFM U-Net [64,128,256,512,1024] 3x{TILE}x{TILE} H&E->IHC tiles, "
                                   f"batch {B}/GPU, sample+fwd+loss+bwd+allreduce+Adam",
                       "global_batch": B * world, "tile": TILE, "parallelism": f"dp{world}",
                       "final_loss": round(float(loss), 6), "loss_bits": float(loss).hex(),
                       "grad_exchange": (trainer.bucketer.mode if trainer.bucketer.enabled else "none"),
                       "sync_batchnorm": trainer._sync_bn is not None,
                       "launch": _launch_note(trainer.graph),
                       ("eager_ms_per_step" if trainer.graph else "graph_ms_per_step"): other_ms,
                       "buckets_mb": [round((hi - lo) * 4 / 2 ** 20, 2) for _, lo, hi in trainer.bucketer.buckets]},
            "roofline": {"bound": "mfma", "kernel": "conv3x3 fwd + dgrad launches (conv3x3_dma16_kernel; conv3x3_stage_kernel on the 256^2 level)",
                         "achieved": round(achieved, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": traffic,
                         "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE*2 + WRITE_SIZE, profiles/hbm_traffic_current.json)",
                         "launches_per_step": n_l // timed_steps, "launches_timed": n_l,
                         "avg_launch_ms": round(t_l * 1e3 / n_l, 4),
                         "algorithmic_gflop_per_launch": round(f_l / n_l / 1e9, 3),
                         "note": "event-bracketed steps (1 in %d) run single-stream so a bracket times its kernel alone; "
                                 "the other steps overlap the weight gradients with the HBM-bound backward passes on a "
                                 "side stream (S2S_WGRAD_STREAM=0 disables it, e.g. for rocprofv3 kernel statistics)" % every},
            "kernels": kernels,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
    rc = rccl_object(dev, world, use_dist)            # (a collective: every rank calls)
    if rank == 0:
        out["rccl"] = rc
    # BASELINE.json's metric is worded on the pix2pix G + D step, which the reference does not contain:
    # that step is timed right after the reference-parity line, under the same contract, and reported under "pix2pix"
    p2p = None
    default_line = args.precision == "bf16" and not args.h2d and not args.breakdown
    extras = {}
    if default_line and world == 1 and not args.no_extras and TILE == 256:
        del trainer, net
        trainer = net = None
        torch.cuda.empty_cache()
        extras["sample"] = sample_leg(dev)
        torch.cuda.empty_cache()
        extras["fp32_parity"] = fp32_parity_leg(dev, B)
        torch.cuda.empty_cache()
        extras["dropin"] = dropin_leg(dev, B, max(5, args.steps // 2), out["ms_per_step"])
        torch.cuda.empty_cache()
        mt = multitask_bench(args, dev, rank, world, use_dist, 512, 8, max(4, args.steps // 2), 2)
        extras["multitask"] = {k: mt[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "dtype", "config", "roofline", "kernels")}
        torch.cuda.empty_cache()
    if not args.no_pix2pix and default_line:
        del trainer, net
        torch.cuda.empty_cache()
        p2p = pix2pix_bench(args, dev, rank, world, use_dist)
    if rank == 0:
        out.update(extras)
        if p2p is not None:
            out["pix2pix"] = {k: p2p[k] for k in ("metric", "value", "unit", "ms_per_step", "dtype", "config", "roofline",
                                                  "generator_conv_frac", "generator_conv", "kernels")}
            if world == 1 and not args.no_cpu_baseline:
                out["pix2pix"]["cpu_baseline"] = pix2pix_cpu_baseline()
        _emit(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


def _emit(line: str) -> None:
    """The one JSON line goes to the process's real stdout (see _quiet_stdout)."""
    os.write(_REAL_STDOUT, (line + "\n").encode())


_REAL_STDOUT = 1


def _quiet_stdout() -> None:
    """Libraries write banners to C stdout (RCCL prints its version block there when the communicator is created):
    point fd 1 at stderr for the life of the process and keep the original for the result line, so that stdout
    carries exactly one line."""
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)


if __name__ == "__main__":
    _quiet_stdout()
    main()
