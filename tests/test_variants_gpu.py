"""Row f4 on the GPU: mask-conditioned network (4 input channels), ROI-weighted MSE, Charbonnier ROI value --
against tests/golden/variants_step.npz (reference SharedEncoder(4) + FlowMatchingDecoder) -- and the build-defined
class-conditional network against the CPU oracle's restatement of the same formula (parity unpinned: the
reference's class-conditional network is the absent third-party torchcfm UNetModel)."""
import pytest
import torch

from conftest import load_golden, relerr, sub
from test_e2e_gpu import check_grads

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-3


@pytest.fixture(scope="module")
def G():
    return load_golden("variants_step.npz")


def build_net(G, precision="fp32", cls=None, **kw):
    from stain2stain_amd import FlowUNet
    net = (cls or FlowUNet)(4 if cls is None else 3, [16, 32], 3, 32, precision=precision, **kw)
    if cls is None:
        net.load_state_dict(sub(G, "init/"))
    return net.to(DEV).train()


@pytest.mark.parametrize("shape", [(1, 1, 5, 7), (4, 3, 64, 64), (16, 3, 256, 256)])
def test_weighted_mse_and_charbonnier_kernels_match_oracle(shape):
    from oracle import unet_oracle as O
    from stain2stain_amd import ops
    B, C, H, W = shape
    g = torch.Generator().manual_seed(H)
    v, u = torch.randn(shape, generator=g), torch.randn(shape, generator=g)
    m = (torch.rand(B, 1, H, W, generator=g) > 0.5).float()
    vr = v.clone().requires_grad_(True)
    ref = O.weighted_mse(vr, u, m, 10.0)
    ref.backward()
    out, dv = ops.weighted_mse(v.to(DEV), u.to(DEV), m.to(DEV), 10.0)
    assert relerr(out[0], ref.detach()) < 1e-5
    assert relerr(dv, vr.grad) < 1e-5
    assert relerr(ops.charbonnier_roi(v.to(DEV), u.to(DEV), m.to(DEV))[0], O.charbonnier_roi(v, u, m)) < 1e-5
    z = torch.zeros_like(m)                                    # empty ROI: plain MSE, Charbonnier value 0
    assert relerr(ops.weighted_mse(v.to(DEV), u.to(DEV), z.to(DEV), 10.0, want_grad=False)[0][0],
                  torch.mean((v - u) ** 2)) < 1e-5
    assert float(ops.charbonnier_roi(v.to(DEV), u.to(DEV), z.to(DEV))[0]) == 0.0
    with pytest.raises(RuntimeError):
        ops.weighted_mse(v.to(DEV), u.to(DEV), torch.zeros(B, 1, H + 1, W, device=DEV))


@pytest.mark.parametrize("tag", ["mse", "roi"])
def test_mask_conditioned_step_matches_golden_fp32(G, tag):
    from stain2stain_amd import MaskConditionedFlowMatchingModule
    from stain2stain_amd.flow_matching import _MSE, _WeightedMSE
    net = build_net(G)
    mod = MaskConditionedFlowMatchingModule(net)
    x0, x1, t, mask = (G[k].to(DEV) for k in ("x0", "x1", "t", "mask"))
    _, xt, ut = mod.flow_matcher.sample_location_and_conditional_flow(x0, x1, t)
    v = mod.forward(t, xt, mask)
    loss = _MSE.apply(v, ut) if tag == "mse" else _WeightedMSE.apply(v, ut, mask, 10.0)
    loss.backward()
    assert relerr(v, G[tag + "/v"]) < TOL and relerr(loss, G[tag + "/loss"]) < TOL
    got = {"encoder." + k: p.grad for k, p in net.encoder.named_parameters()}
    got.update({"flow_decoder." + k: p.grad for k, p in net.flow_decoder.named_parameters()})
    check_grads(got, sub(G, tag + "/grad/"), TOL)


def test_variant_modules_model_step(G):
    """The four module variants run their reference-shaped batches; toggling and the ROI terms act as documented."""
    from stain2stain_amd import (FlowUNet, MaskConditionedFlowMatchingModule, ROICharbonnierFlowMatchingModule,
                                 ROIWeightedFlowMatchingModule)
    from stain2stain_amd import ops
    x0, x1, mask = (G[k].to(DEV) for k in ("x0", "x1", "mask"))
    net3 = FlowUNet(3, [16, 32], 3, 32, precision="fp32").to(DEV).train()
    torch.manual_seed(3)
    lw = ROIWeightedFlowMatchingModule(net3).model_step((x0, x1, mask))
    torch.manual_seed(3)
    lc = ROICharbonnierFlowMatchingModule(net3).model_step((x0, x1, mask))
    torch.manual_seed(3)
    t = torch.rand(4, device=DEV)
    xt, ut = ops.cfm_sample(x0, x1, t, 0.0, None)
    charb = ops.charbonnier_roi(xt, x1, mask)[0]
    assert float(charb) > 0 and torch.isfinite(lw) and float(lc.detach()) > float(charb)
    lw.backward()
    assert all(p.grad is not None for p in net3.parameters())
    from stain2stain_amd import SolverConfig
    mod = MaskConditionedFlowMatchingModule(build_net(G), mask_toggle=True, solver=SolverConfig("euler"))
    seen = set()
    for s in range(8):                                             # torch.rand(1) < 0.5 decides, as in the reference
        torch.manual_seed(s)
        toggled = torch.rand(1).item() < 0.5
        torch.manual_seed(s)
        loss = mod.training_step((x0, x1, mask), 0)
        seen.add(toggled)
        assert torch.isfinite(loss)
    assert seen == {True, False}
    img = mod.generate(x0[:2], mask[:2], num_steps=2)
    assert img.shape == (2, 3, 64, 64) and not mod.net.training      # eval mode stays, as in the reference


def test_class_conditional_net_matches_oracle_fp32(G):
    from oracle import unet_oracle as O
    from stain2stain_amd import ClassConditionalFlowMatchingModule, ClassConditionalFlowUNet
    x0, x1, t = (G[k] for k in ("x0", "x1", "t"))
    y = torch.tensor([2, 0, 2, 1])
    # initialisation screened like the golden draws (make_golden.py): the first seed whose oracle gradients do not
    # move under 1e-6 input jitter, i.e. no ReLU / max-pool decision sits on a knife edge (most seeds fail this:
    # measured 10 of 13 move their gradients by 2e-3 .. 1.6e-2)
    for seed in range(7, 60):
        torch.manual_seed(seed)
        net = ClassConditionalFlowUNet(3, [16, 32], 3, 32, num_classes=3, precision="fp32")
        P = {k: v.detach().clone() for k, v in net.state_dict().items()}
        base = O.variant_loss_and_grads(P, x0, x1, t, y=y)[2]
        scale = max(float(v.abs().max()) for v in base.values())
        worst = 0.0
        for j in (1, 2, 3):
            gj = torch.Generator().manual_seed(j)
            other = O.variant_loss_and_grads(P, x0 + 1e-6 * torch.randn(x0.shape, generator=gj),
                                             x1 + 1e-6 * torch.randn(x1.shape, generator=gj), t, y=y)[2]
            worst = max(worst, max(float((other[k] - base[k]).abs().max())
                                   / max(float(base[k].abs().max()), 1e-3 * scale) for k in base))
        if worst < 3e-4:
            break
    else:
        pytest.fail("no well-conditioned initialisation found")
    net = net.to(DEV).train()
    from stain2stain_amd import SolverConfig
    mod = ClassConditionalFlowMatchingModule(net, solver=SolverConfig("euler"))
    _, xt, ut = mod.flow_matcher.sample_location_and_conditional_flow(x0.to(DEV), x1.to(DEV), t.to(DEV))
    v = mod.forward(t.to(DEV), xt, y.to(DEV))
    loss = torch.mean((v - ut) ** 2)
    loss.backward()
    rl, rv, rg, _ = O.variant_loss_and_grads(P, x0, x1, t, y=y)
    assert relerr(v, rv) < TOL and relerr(loss, rl) < TOL
    got = {k: p.grad for k, p in net.named_parameters()}
    check_grads(got, rg, TOL)
    assert float(got["label_emb.weight"][1].abs().max()) > 0
    # y=None is the unconditional network; out-of-range labels raise like nn.Embedding
    assert relerr(net(t.to(DEV), xt), net(t.to(DEV), xt, y=None)) == 0
    with pytest.raises(IndexError):
        net(t.to(DEV), xt, y=torch.tensor([0, 1, 2, 3], device=DEV))
    img = mod.generate(x0[:2].to(DEV), 1, num_steps=2)
    assert img.shape == (2, 3, 64, 64)
