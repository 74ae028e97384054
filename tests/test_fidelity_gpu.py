"""Two claims of DESIGN.md turned into tests.

1. The bf16 throughput mode TRAINS like the fp32 parity mode: same network, same initialisation, same stream of fresh
   batches, 160 optimisation steps each; the smoothed loss curves and a held-out evaluation loss of the two runs stay
   within a stated band.  Reference semantics of the step: src/models/conditional_flow_matching.py:53-88.
2. BASELINE.json configs[4] -- the multi-task / any2any models on 512x512 tiles, batch 8 per GPU -- runs at that size:
   forward of both heads against the CPU oracle in fp32 mode at 1e-3 on two samples, and the full batch-8 training steps
   of the multitask and the class-conditional modules are finite and learn.
"""
import os

import pytest
import torch

from conftest import relerr

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _stain_pair(gen, b, hw):
    """A synthetic paired H&E -> IHC batch with structure to learn: smooth random fields (the 'tissue') and a fixed
    colour unmixing + tone curve as the 'stain' map.  Fresh fields on every call."""
    low = torch.rand(b, 3, hw // 8, hw // 8, generator=gen)
    x0 = torch.nn.functional.interpolate(low, size=(hw, hw), mode="bilinear", align_corners=False) * 2 - 1
    x0 = x0 + 0.1 * (torch.rand(b, 3, hw, hw, generator=gen) - 0.5)
    mix = torch.tensor([[0.6, 0.3, 0.1], [0.1, 0.7, 0.2], [0.25, 0.05, 0.7]])
    x1 = torch.tanh(1.5 * torch.einsum("oc,bchw->bohw", mix, x0)) * 0.9
    return x0.clamp(-1, 1).contiguous(), x1.contiguous()


def test_bf16_mode_trains_like_the_fp32_mode():
    from stain2stain_amd import CFMTrainer, FlowUNet
    steps, B, HW = 160, 8, 64
    gen_h = torch.Generator().manual_seed(99)
    x0h, x1h = _stain_pair(gen_h, 16, HW)
    th = torch.rand(16, generator=gen_h)
    curves, held = {}, {}
    for prec in ("fp32", "bf16"):
        torch.manual_seed(2024)
        net = FlowUNet(3, [32, 64, 128], 3, 64, precision=prec).to(DEV).train()
        tr = CFMTrainer(net, lr=1e-3, weight_decay=1e-5)
        gen = torch.Generator().manual_seed(7)                     # the same stream of fresh batches for both runs
        losses = []
        for _ in range(steps):
            x0, x1 = _stain_pair(gen, B, HW)
            t = torch.rand(B, generator=gen)
            losses.append(tr.step(x0.to(DEV), x1.to(DEV), t.to(DEV)))
        curves[prec] = torch.stack(losses).float().cpu()
        net.eval()
        with torch.no_grad():
            tb = th.view(-1, 1, 1, 1)
            xt, ut = tb * x1h + (1 - tb) * x0h, x1h - x0h
            v = net(th.to(DEV), xt.to(DEV)).float().cpu()
        held[prec] = float(((v - ut) ** 2).mean())
    smooth = {k: torch.nn.functional.avg_pool1d(c[None, None], 16, 8)[0, 0] for k, c in curves.items()}
    band = float(((smooth["bf16"] - smooth["fp32"]).abs() / smooth["fp32"]).max())
    drop = {k: float(c[-16:].mean() / c[:4].mean()) for k, c in curves.items()}
    print(f"smoothed train-loss curves differ by at most {band:.3f} (relative); final/initial loss fp32 {drop['fp32']:.3f}, "
          f"bf16 {drop['bf16']:.3f}; held-out eval loss fp32 {held['fp32']:.5f}, bf16 {held['bf16']:.5f}")
    assert drop["fp32"] < 0.35 and drop["bf16"] < 0.35                  # both runs learn the mapping
    assert band < 0.10                                                  # curves within 10 % of each other, window by window
    # and generalise alike: the bf16 run may not be more than 15 % worse on held-out data (nor implausibly better)
    assert held["bf16"] < 1.15 * held["fp32"] and held["bf16"] > 0.7 * held["fp32"]


def test_bf16_mode_trains_like_the_fp32_mode_at_production_width():
    """The same claim on the network the bench times (at production widths the encoder's bf16 gradient cosine
    to fp32 is 0.84-0.87, and only a [32,64,128] net had been trained in both modes): features [64,...,1024], batch 16,
    256x256, 200 optimisation steps per mode from one initialisation on one stream of batches."""
    from stain2stain_amd import CFMTrainer, FlowUNet
    steps, B, HW, POOL = 200, 16, 256, 10
    gen = torch.Generator().manual_seed(11)
    pool = [tuple(t.to(DEV) for t in _stain_pair(gen, B, HW)) for _ in range(POOL)]
    ts = torch.rand(steps, B, generator=gen).to(DEV)
    x0h, x1h = (t.to(DEV) for t in _stain_pair(gen, B, HW))
    th = torch.rand(B, generator=gen).to(DEV)
    curves, held = {}, {}
    for prec in ("fp32", "bf16"):
        torch.manual_seed(2024)
        net = FlowUNet(3, [64, 128, 256, 512, 1024], 3, 256, precision=prec).to(DEV).train()
        tr = CFMTrainer(net, lr=2e-4, weight_decay=1e-5)
        losses = []
        for i in range(steps):
            x0, x1 = pool[i % POOL]
            losses.append(tr.step(x0, x1, ts[i]))
        curves[prec] = torch.stack(losses).float().cpu()
        net.eval()
        with torch.no_grad():
            tb = th.view(-1, 1, 1, 1)
            v = net(th, tb * x1h + (1 - tb) * x0h).float()
        held[prec] = float(((v - (x1h - x0h)) ** 2).mean())
        del tr, net
        torch.cuda.empty_cache()
    smooth = {k: torch.nn.functional.avg_pool1d(c[None, None], 20, 10)[0, 0] for k, c in curves.items()}
    band = float(((smooth["bf16"] - smooth["fp32"]).abs() / smooth["fp32"]).max())
    drop = {k: float(c[-20:].mean() / c[:4].mean()) for k, c in curves.items()}
    print(f"production width: smoothed train-loss curves differ by at most {band:.3f} (relative); final/initial loss fp32 "
          f"{drop['fp32']:.3f}, bf16 {drop['bf16']:.3f}; held-out eval loss fp32 {held['fp32']:.5f}, bf16 {held['bf16']:.5f}")
    assert drop["fp32"] < 0.35 and drop["bf16"] < 0.35
    assert band < 0.15
    assert abs(held["bf16"] - held["fp32"]) < 0.25 * held["fp32"]


def test_config4_multitask_and_class_conditional_at_512_batch_8():
    from oracle import unet_oracle as O
    from stain2stain_amd import (ClassConditionalFlowMatchingModule, ClassConditionalFlowUNet, FlowMatchingDecoder,
                                 MultiTaskFlowMatchingModule, SegmentationDecoder, SharedEncoder)
    feats = [64, 128, 256, 512, 1024]
    g = torch.Generator().manual_seed(512)
    B, HW = 8, 512
    x0 = torch.rand(B, 3, HW, HW, generator=g) * 2 - 1
    x1 = torch.rand(B, 3, HW, HW, generator=g) * 2 - 1
    t = torch.rand(B, generator=g)
    mask = (torch.rand(B, 1, HW, HW, generator=g) > 0.5).float()
    # ---- forward of both heads on two samples, fp32 mode vs the CPU oracle (1e-3) ----
    torch.manual_seed(1984)
    mod = MultiTaskFlowMatchingModule(SharedEncoder(3, feats, precision="fp32"),
                                      FlowMatchingDecoder(feats[-1], feats[:-1][::-1], 3, 256, precision="fp32"),
                                      SegmentationDecoder(feats[-1], feats[:-1][::-1], 1, precision="fp32"))
    P = {"encoder." + k: v.clone() for k, v in mod.encoder.state_dict().items()}
    P.update({"flow_decoder." + k: v.clone() for k, v in mod.flow_decoder.state_dict().items()})
    P.update({"seg_decoder." + k: v.clone() for k, v in mod.seg_decoder.state_dict().items()})
    mod = mod.to(DEV).train()
    tb = t[:2].view(-1, 1, 1, 1)
    xt = tb * x1[:2] + (1 - tb) * x0[:2]
    with torch.no_grad():
        v = mod.forward_flow(t[:2].to(DEV), xt.to(DEV)).float().cpu()
        z = mod.forward_segmentation(x0[:2].to(DEV)).float().cpu()
    threads = torch.get_num_threads()
    torch.set_num_threads(min(16, os.cpu_count() or 8))
    try:
        with torch.no_grad():
            v_ref = O.flow_forward(t[:2], xt, P, True)
            b_, skips = O.encoder_forward(x0[:2], P, True)
            z_ref = O.seg_decoder_forward(b_, skips, P, True)
    finally:
        torch.set_num_threads(threads)
    print(f"512x512 fp32 mode: velocity {relerr(v, v_ref):.2e}, mask logits {relerr(z, z_ref):.2e}")
    assert relerr(v, v_ref) < 1e-3 and relerr(z, z_ref) < 1e-3
    del mod
    torch.cuda.empty_cache()
    # ---- the full batch of 8 in the throughput mode: multitask step (flow + Dice/BCE, encoder run twice) learns ----
    torch.manual_seed(1984)
    mod = MultiTaskFlowMatchingModule(SharedEncoder(3, feats), FlowMatchingDecoder(feats[-1], feats[:-1][::-1], 3, 256),
                                      SegmentationDecoder(feats[-1], feats[:-1][::-1], 1)).to(DEV).train()
    opt = torch.optim.Adam(list(mod.encoder.parameters()) + list(mod.flow_decoder.parameters())
                           + list(mod.seg_decoder.parameters()), lr=1e-3)
    batch = (x0.to(DEV), x1.to(DEV), mask.to(DEV))
    hist = []
    for _ in range(3):
        opt.zero_grad()
        torch.manual_seed(3)                               # the same t every step, so the loss is comparable
        total, d = mod.model_step(batch)
        total.backward()
        opt.step()
        hist.append(float(total.detach()))
    assert all(h == h for h in hist) and hist[-1] < hist[0], hist
    assert all(torch.isfinite(p.grad).all() for p in mod.seg_decoder.parameters())
    del mod, opt
    torch.cuda.empty_cache()
    # ---- class-conditional (any2any) module: 4 stain domains, batch 8 at 512x512 ----
    torch.manual_seed(1984)
    net = ClassConditionalFlowUNet(3, feats, 3, 256, num_classes=4).to(DEV).train()
    cc = ClassConditionalFlowMatchingModule(net)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    y = torch.randint(0, 4, (B,), generator=g).to(DEV)
    hist = []
    for _ in range(3):
        opt.zero_grad()
        torch.manual_seed(3)
        loss = cc.model_step((x0.to(DEV), x1.to(DEV), y))
        loss.backward()
        opt.step()
        hist.append(float(loss.detach()))
    assert all(h == h for h in hist) and hist[-1] < hist[0], hist
    assert float(net.label_emb.weight.grad.abs().max()) > 0
