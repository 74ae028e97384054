"""Row f2 on the GPU: mask head (SegmentationDecoder), Dice + BCE loss kernel and the multitask step, against
tests/golden/multitask_step.npz (reference classes) and the CPU oracle.  Tolerance 1e-3 relative in fp32 mode."""
import pytest
import torch

from conftest import load_golden, relerr, sub
from test_e2e_gpu import check_grads

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-3


@pytest.fixture(scope="module")
def G():
    return load_golden("multitask_step.npz")


def build(G, precision):
    from stain2stain_amd import (FlowMatchingDecoder, MultiTaskFlowMatchingModule, SegmentationDecoder,
                                 SharedEncoder)
    enc = SharedEncoder(3, [16, 32], precision=precision)
    fdec = FlowMatchingDecoder(32, [16], 3, 32, precision=precision)
    sdec = SegmentationDecoder(32, [16], 1, precision=precision)
    enc.load_state_dict(sub(G, "init/encoder."))
    fdec.load_state_dict(sub(G, "init/flow_decoder."))
    sdec.load_state_dict(sub(G, "init/seg_decoder."))
    from stain2stain_amd import SolverConfig
    return MultiTaskFlowMatchingModule(enc, fdec, sdec, time_emb_dim=32, solver=SolverConfig()).to(DEV).train()


@pytest.mark.parametrize("n", [1, 777, 4 * 64 * 64, 16 * 256 * 256])
@pytest.mark.parametrize("dw", [0.5, 0.0, 1.0])
def test_seg_loss_kernel_matches_oracle(n, dw):
    from oracle import unet_oracle as O
    from stain2stain_amd import ops
    g = torch.Generator().manual_seed(n)
    z = torch.randn(n, generator=g) * 3
    m = (torch.rand(n, generator=g) > 0.6).float()
    zr = z.clone().requires_grad_(True)
    seg, d, b = O.seg_loss(zr, m, dw)
    seg.backward()
    out, dz = ops.seg_loss(z.to(DEV), m.to(DEV), 1.0, dw, want_grad=True)
    assert relerr(out[0], seg.detach()) < 1e-5 and relerr(out[1], d.detach()) < 1e-5
    assert relerr(out[2], b.detach()) < 1e-5
    assert float((dz.cpu() - zr.grad).abs().max()) <= 1e-5 * float(zr.grad.abs().max()) + 1e-12


def test_seg_loss_edge_cases():
    from stain2stain_amd import ops
    z = torch.tensor([80.0, -80.0, 0.0, 30.0], device=DEV)          # saturated logits stay finite
    out, dz = ops.seg_loss(z, torch.tensor([0.0, 1.0, 1.0, 1.0], device=DEV), 1.0, 0.5, want_grad=True)
    assert torch.isfinite(out).all() and torch.isfinite(dz).all()
    out, _ = ops.seg_loss(torch.full((64,), -50.0, device=DEV), torch.zeros(64, device=DEV), 1.0, 1.0)
    assert abs(float(out[1])) < 1e-6                                 # empty mask, empty prediction: Dice loss 0
    with pytest.raises(RuntimeError):
        ops.seg_loss(torch.zeros(4), torch.zeros(4), 1.0, 0.5)       # host tensors are refused


def test_seg_decoder_state_dict_keys_match_reference(G):
    from stain2stain_amd import SegmentationDecoder
    sd = SegmentationDecoder(32, [16], 1).state_dict()
    ref = sub(G, "init/seg_decoder.")
    assert list(sd.keys()) == list(ref.keys())
    assert all(tuple(sd[k].shape) == tuple(ref[k].shape) for k in ref)


def test_multitask_step_matches_golden_fp32(G):
    mod = build(G, "fp32")
    x0, x1, t, mask = (G[k].to(DEV) for k in ("x0", "x1", "t", "mask"))
    _, xt, ut = mod.flow_matcher.sample_location_and_conditional_flow(x0, x1, t)
    flow = torch.mean((mod.forward_flow(t, xt) - ut) ** 2)
    logits = mod.forward_segmentation(x0)
    assert relerr(logits, G["logits"]) < TOL
    seg, d = mod.compute_segmentation_loss(logits, mask)
    total = flow + 1.0 * seg
    total.backward()
    for k, v in (("total", total), ("flow", flow), ("dice", d["dice"]), ("bce", d["bce"])):
        assert relerr(v, G["loss/" + k]) < TOL, k
    got = {}
    for pre, m in (("encoder.", mod.encoder), ("flow_decoder.", mod.flow_decoder), ("seg_decoder.", mod.seg_decoder)):
        got.update({pre + k: p.grad for k, p in m.named_parameters()})
    check_grads(got, sub(G, "grad/"), TOL)
    after = sub(G, "after/")
    for pre, m in (("encoder.", mod.encoder), ("seg_decoder.", mod.seg_decoder)):
        for k, v in m.state_dict().items():
            r = after[pre + k]
            if k.endswith("num_batches_tracked"):
                assert int(v) == int(r), k
            elif "running" in k:
                assert relerr(v, r) < TOL, k


def test_model_step_and_generate(G):
    """model_step (own t draw) returns the loss dictionary of the reference; bf16 mode stays near fp32."""
    x0, x1, mask = (G[k].to(DEV) for k in ("x0", "x1", "mask"))
    vals = {}
    for prec in ("fp32", "bf16"):
        mod = build(G, prec)
        torch.manual_seed(5)
        total, d = mod.model_step((x0, x1, mask))
        assert set(d) == {"total", "flow", "seg", "seg_dice", "seg_bce"}
        total.backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in mod.seg_decoder.parameters())
        vals[prec] = float(total.detach())
    assert abs(vals["bf16"] - vals["fp32"]) < 2e-2 * abs(vals["fp32"])
    mod = build(G, "fp32")
    # generate(): eval mode during and after the call (the reference's self.eval(), :437), BatchNorm running
    # statistics untouched, the solver's adaptive dopri5 by default and fixed-step Euler on request
    mod.train()
    before = {k: v.clone() for k, v in mod.state_dict().items() if "running_" in k or "num_batches" in k}
    img, pm = mod.generate(x0[:2], num_steps=3, method="euler")
    assert img.shape == (2, 3, 64, 64) and pm.shape == (2, 1, 64, 64)
    assert float(pm.min()) >= 0 and float(pm.max()) <= 1
    assert not mod.training and not any(m.training for m in mod.modules())
    img2, _ = mod.generate(x0[:2])                                   # dopri5, atol = rtol = 1e-4
    fine, _ = mod.generate(x0[:2], num_steps=200, method="euler")
    assert relerr(img2, fine) < 2e-2 and not mod.training
    after = mod.state_dict()
    assert before and all(torch.equal(after[k], v) for k, v in before.items())
    mod.solver = None
    with pytest.raises(ValueError, match="Solver is not initialized"):
        mod.generate(x0[:2])


# ---- multiclass form ------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def GM():
    return load_golden("multiclass_step.npz")


def test_multiclass_loss_kernel_matches_golden(GM):
    from stain2stain_amd import ops
    out, dz = ops.seg_loss_multiclass(GM["lossop/z"].to(DEV), GM["lossop/target"].to(DEV), ignore_index=2,
                                      dice_weight=0.3)
    assert relerr(out[1], GM["lossop/dice"]) < 1e-5 and relerr(out[2], GM["lossop/ce"]) < 1e-5
    assert relerr(dz, GM["lossop/dz"]) < 1e-4


@pytest.mark.parametrize("C", [2, 3, 5, 8])
@pytest.mark.parametrize("ignore", [-100, 1])
def test_multiclass_loss_kernel_matches_oracle(C, ignore):
    from oracle import unet_oracle as O
    from stain2stain_amd import ops
    g = torch.Generator().manual_seed(C)
    z = torch.randn(3, C, 37, 53, generator=g) * 4
    t = torch.randint(0, C, (3, 37, 53), generator=g)
    zr = z.clone().requires_grad_(True)
    seg, d, ce = O.seg_loss_multiclass(zr, t, 0.5, ignore)
    seg.backward()
    out, dz = ops.seg_loss_multiclass(z.to(DEV), t.to(DEV), ignore)
    assert relerr(out[0], seg.detach()) < 1e-5 and relerr(out[1], d.detach()) < 1e-5
    assert relerr(out[2], ce.detach()) < 1e-5
    assert relerr(dz, zr.grad) < 1e-4


def test_multiclass_loss_rejects_bad_input():
    from stain2stain_amd import ops
    z = torch.zeros(1, 3, 4, 4, device=DEV)
    with pytest.raises(RuntimeError):                      # label 3 with 3 classes: F.one_hot raises in the reference
        ops.seg_loss_multiclass(z, torch.full((1, 4, 4), 3, device=DEV))
    with pytest.raises(RuntimeError):
        ops.seg_loss_multiclass(z, torch.zeros(1, 4, 5, dtype=torch.long, device=DEV))
    with pytest.raises(RuntimeError):
        ops.seg_loss_multiclass(torch.zeros(1, 9, 4, 4, device=DEV), torch.zeros(1, 4, 4, dtype=torch.long, device=DEV))


def test_multiclass_step_matches_golden_fp32(GM):
    from stain2stain_amd import (FlowMatchingDecoder, MultiTaskFlowMatchingModule, SegmentationDecoder,
                                 SharedEncoder)
    G = GM
    enc = SharedEncoder(3, [16, 32], precision="fp32")
    fdec = FlowMatchingDecoder(32, [16], 3, 32, precision="fp32")
    sdec = SegmentationDecoder(32, [16], 5, precision="fp32")
    enc.load_state_dict(sub(G, "init/encoder."))
    fdec.load_state_dict(sub(G, "init/flow_decoder."))
    sdec.load_state_dict(sub(G, "init/seg_decoder."))
    from stain2stain_amd import SolverConfig
    mod = MultiTaskFlowMatchingModule(enc, fdec, sdec, time_emb_dim=32, num_classes=5,
                                      solver=SolverConfig("euler")).to(DEV).train()
    x0, x1, t, mask = (G[k].to(DEV) for k in ("x0", "x1", "t", "mask"))
    _, xt, ut = mod.flow_matcher.sample_location_and_conditional_flow(x0, x1, t)
    flow = torch.mean((mod.forward_flow(t, xt) - ut) ** 2)
    logits = mod.forward_segmentation(x0)
    assert relerr(logits, G["logits"]) < TOL
    seg, d = mod.compute_segmentation_loss(logits, mask.unsqueeze(1))     # (B,1,H,W) masks are squeezed (:224-225)
    total = flow + seg
    total.backward()
    for k, v in (("total", total), ("flow", flow), ("dice", d["dice"]), ("ce", d["ce"])):
        assert relerr(v, G["loss/" + k]) < TOL, k
    got = {}
    for pre, m in (("encoder.", mod.encoder), ("flow_decoder.", mod.flow_decoder), ("seg_decoder.", mod.seg_decoder)):
        got.update({pre + k: p.grad for k, p in m.named_parameters()})
    check_grads(got, sub(G, "grad/"), TOL)
    img, cls = mod.generate(x0[:1], num_steps=2)
    assert cls.dtype == torch.int64 and cls.shape == (1, 1, 64, 64) and int(cls.max()) < 5
