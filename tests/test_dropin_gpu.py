"""The drop-in path's optimiser (stain2stain_amd.FusedAdam: torch.optim.Adam's update as one HIP launch over ordinary
module parameters) and the modules under autograd with weight gradients on the side stream."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(64, 3, 3, 3), (64,), (128, 64, 3, 3), (5000,), (7,), (256, 256)]
    return [torch.randn(s, generator=g).to(DEV).requires_grad_(True) for s in shapes]


def test_fused_adam_matches_torch_adam_and_shares_its_state_dict():
    from stain2stain_amd import FusedAdam
    pa, pb = _params(1), _params(1)
    oa = FusedAdam(pa, lr=1e-3, weight_decay=1e-2)
    ob = torch.optim.Adam(pb, lr=1e-3, weight_decay=1e-2, foreach=False)
    g = torch.Generator().manual_seed(2)
    for step in range(4):
        for a, b in zip(pa, pb):
            gr = torch.randn(a.shape, generator=g).to(DEV)
            a.grad, b.grad = gr.clone(), gr.clone()
        if step == 2:                     # a parameter without a gradient is skipped, its step count stays behind
            pa[3].grad = pb[3].grad = None
        v0 = pa[0]._version
        oa.step(); ob.step()
        assert pa[0]._version > v0        # packed-weight caches see the update
    for a, b in zip(pa, pb):
        assert float((a.detach() - b.detach()).abs().max() / b.detach().abs().max()) < 2e-6
    sa, sb = oa.state_dict(), ob.state_dict()
    assert sa["param_groups"][0]["params"] == sb["param_groups"][0]["params"]
    for i in sb["state"]:
        assert float(sa["state"][i]["step"]) == float(sb["state"][i]["step"])
        assert float((sa["state"][i]["exp_avg"] - sb["state"][i]["exp_avg"]).abs().max()) < 1e-6
        assert float((sa["state"][i]["exp_avg_sq"] - sb["state"][i]["exp_avg_sq"]).abs().max()) < 1e-6
    # the other class resumes from it
    oc = torch.optim.Adam(_params(1), lr=1e-3, weight_decay=1e-2)
    oc.load_state_dict(sa)
    od = FusedAdam(_params(1), lr=1e-3, weight_decay=1e-2)
    od.load_state_dict(sb)
    for p in od.param_groups[0]["params"]:
        p.grad = torch.ones_like(p)
    od.step()


def test_fused_adam_refuses_cpu_parameters():
    from stain2stain_amd import FusedAdam
    p = torch.zeros(4, requires_grad=True)
    p.grad = torch.ones(4)
    with pytest.raises(RuntimeError, match="GPU"):
        FusedAdam([p]).step()


def test_dropin_training_step_with_fused_adam_tracks_the_fused_trainer():
    """zero_grad / training_step / backward / FusedAdam.step on the modules (weight gradients on the side stream inside
    the autograd backward) against CFMTrainer on the same batch and t: same kernels except the loss head and the optimiser
    launch, so the parameters agree to rounding after two steps."""
    from functools import partial
    from stain2stain_amd import CFMTrainer, ConditionalFlowMatcher, FlowUNet, FusedAdam
    g = torch.Generator().manual_seed(5)
    x0 = (torch.rand(4, 3, 64, 64, generator=g) * 2 - 1).to(DEV)
    x1 = (torch.rand(4, 3, 64, 64, generator=g) * 2 - 1).to(DEV)
    ts = [torch.rand(4, generator=g).to(DEV) for _ in range(2)]
    torch.manual_seed(3)
    na = FlowUNet(3, [16, 32, 64], 3, 32, precision="fp32").to(DEV).train()
    torch.manual_seed(3)
    nb = FlowUNet(3, [16, 32, 64], 3, 32, precision="fp32").to(DEV).train()
    opt = FusedAdam(list(na.parameters()), lr=1e-3, weight_decay=1e-5)
    fm = ConditionalFlowMatcher(0.0)
    tr = CFMTrainer(nb, lr=1e-3, weight_decay=1e-5)
    for s, t in enumerate(ts):
        opt.zero_grad()
        _, xt, ut = fm.sample_location_and_conditional_flow(x0, x1, t)
        loss = torch.mean((na(t, xt) - ut) ** 2)
        loss.backward()
        lb, _ = tr.forward_backward(x0, x1, t)
        assert abs(float(loss) - float(lb)) < (1e-5 if s == 0 else 2e-3) * abs(float(lb))
        if s == 0:      # identical parameters: the two backward passes agree to rounding (head + loss are fused in one only)
            for (k, a), (_, b) in zip(na.named_parameters(), nb.named_parameters()):
                scale = float(b.grad.abs().max())
                assert float((a.grad - b.grad).abs().max()) <= 1e-4 * scale + 1e-9, k
        opt.step()
        tr.optimizer_step()
    # (after Adam, elements whose gradient is rounding noise move by +-lr in either run: compare the bulk)
    for (k, a), (_, b) in zip(na.named_parameters(), nb.named_parameters()):
        assert float((a - b).norm()) <= 2e-3 * float(b.norm()) + 1e-6, k
