"""The fused multitask trainer (row f2 / BASELINE.json configs[4]: shared encoder run twice, flow head, mask head, Dice +
BCE, one flat Adam) against tests/golden/multitask_step.npz -- the reference's own classes -- at 1e-3 in fp32 mode, and
against the module path it replaces."""
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


def parts(G, precision, nseg=1, init=None):
    from stain2stain_amd import FlowMatchingDecoder, SegmentationDecoder, SharedEncoder
    enc = SharedEncoder(3, [16, 32], precision=precision)
    fdec = FlowMatchingDecoder(32, [16], 3, 32, precision=precision)
    sdec = SegmentationDecoder(32, [16], nseg, precision=precision)
    enc.load_state_dict(sub(G, "init/encoder."))
    fdec.load_state_dict(sub(G, "init/flow_decoder."))
    sdec.load_state_dict(sub(G, "init/seg_decoder."))
    return enc.to(DEV).train(), fdec.to(DEV).train(), sdec.to(DEV).train()


def test_fused_multitask_step_matches_golden_fp32(G):
    from stain2stain_amd import MultiTaskTrainer
    enc, fdec, sdec = parts(G, "fp32")
    lr, wd = 1e-4, 1e-5
    tr = MultiTaskTrainer(enc, fdec, sdec, time_emb_dim=32, lr=lr, weight_decay=wd, seg_loss_weight=1.0, dice_weight=0.5)
    x0, x1, t, mask = (G[k].to(DEV) for k in ("x0", "x1", "t", "mask"))
    losses, outs = tr.forward_backward(x0, x1, mask, t, want_outputs=True)
    assert relerr(outs["logits"], G["logits"]) < TOL
    for k, mine in (("total", "total"), ("flow", "flow"), ("dice", "seg_dice"), ("bce", "seg_bce")):
        assert relerr(losses[mine], G["loss/" + k]) < TOL, k
    mods = (("encoder.", enc), ("flow_decoder.", fdec), ("seg_decoder.", sdec))
    got = {}
    for pre, m in mods:
        got.update({pre + k: p.grad.clone() for k, p in m.named_parameters()})
    check_grads(got, sub(G, "grad/"), TOL)
    after = sub(G, "after/")
    for pre, m in mods:                   # BatchNorm running statistics: the encoder has seen two batches
        for k, v in m.state_dict().items():
            if k.endswith("num_batches_tracked"):
                assert int(v) == int(after[pre + k]), k
            elif "running" in k:
                assert relerr(v, after[pre + k]) < TOL, k
    # the Adam step: torch.optim.Adam on the reference's own initial weights and gradients (CPU)
    ref_p = {k: G["init/" + k].clone().requires_grad_(True) for k in got}
    for k, p in ref_p.items():
        p.grad = G["grad/" + k].clone()
    torch.optim.Adam(list(ref_p.values()), lr=lr, weight_decay=wd).step()
    tr.optimizer_step()
    assert tr.step_count == 1
    for pre, m in mods:
        for k, p in m.named_parameters():
            r = ref_p[pre + k].detach()
            slack = 1.0 if k.endswith(("double_conv.0.bias", "double_conv.3.bias")) else 0.05    # (zero-gradient biases)
            assert float((p.detach().cpu() - r).abs().max()) <= TOL * float(r.abs().max()) + slack * lr, pre + k


def test_fused_trainer_equals_the_module_path_and_trains(G):
    """Same initial weights, batch and t: the fused step's losses and gradients equal the module path's (autograd over the
    same engine passes) to rounding; a few bf16 steps on a fixed batch lower the loss."""
    from stain2stain_amd import MultiTaskFlowMatchingModule, MultiTaskTrainer, SolverConfig
    x0, x1, t, mask = (G[k].to(DEV) for k in ("x0", "x1", "t", "mask"))
    enc, fdec, sdec = parts(G, "fp32")
    mod = MultiTaskFlowMatchingModule(enc, fdec, sdec, time_emb_dim=32, solver=SolverConfig(), seg_loss_weight=0.7,
                                      dice_weight=0.3).to(DEV).train()
    _, xt, ut = mod.flow_matcher.sample_location_and_conditional_flow(x0, x1, t)
    flow = torch.mean((mod.forward_flow(t, xt) - ut) ** 2)
    seg, _ = mod.compute_segmentation_loss(mod.forward_segmentation(x0), mask)
    (flow + 0.7 * seg).backward()
    want = {k: p.grad.clone() for k, p in mod.named_parameters()}
    e2, f2, s2 = parts(G, "fp32")
    tr = MultiTaskTrainer(e2, f2, s2, time_emb_dim=32, seg_loss_weight=0.7, dice_weight=0.3)
    losses, _ = tr.forward_backward(x0, x1, mask, t)
    assert relerr(losses["total"], (flow + 0.7 * seg).detach()) < 1e-5
    scale = max(float(v.abs().max()) for v in want.values())
    for pre, m in (("encoder.", e2), ("flow_decoder.", f2), ("seg_decoder.", s2)):
        for k, p in m.named_parameters():
            w = want[pre + k]
            assert float((p.grad - w).abs().max()) <= 1e-4 * max(float(w.abs().max()), 1e-3 * scale), pre + k
    e3, f3, s3 = parts(G, "bf16")
    tb = MultiTaskTrainer(e3, f3, s3, time_emb_dim=32, lr=1e-3)
    hist = [float(tb.step(x0, x1, mask, t)) for _ in range(8)]
    assert all(h == h for h in hist) and hist[-1] < hist[0]


def test_multiclass_form_matches_its_golden_step():
    from stain2stain_amd import FlowMatchingDecoder, MultiTaskTrainer, SegmentationDecoder, SharedEncoder
    GM = load_golden("multiclass_step.npz")
    enc = SharedEncoder(3, [16, 32], precision="fp32")
    fdec = FlowMatchingDecoder(32, [16], 3, 32, precision="fp32")
    sdec = SegmentationDecoder(32, [16], 5, precision="fp32")
    enc.load_state_dict(sub(GM, "init/encoder.")); fdec.load_state_dict(sub(GM, "init/flow_decoder."))
    sdec.load_state_dict(sub(GM, "init/seg_decoder."))
    tr = MultiTaskTrainer(enc.to(DEV).train(), fdec.to(DEV).train(), sdec.to(DEV).train(), time_emb_dim=32, num_classes=5,
                          dice_weight=float(GM["meta/dice_weight"]) if "meta/dice_weight" in GM else 0.5,
                          seg_loss_weight=float(GM["meta/seg_loss_weight"]) if "meta/seg_loss_weight" in GM else 1.0,
                          ignore_index=int(GM["meta/ignore_index"]) if "meta/ignore_index" in GM else -100)
    x0, x1, t, mask = (GM[k].to(DEV) for k in ("x0", "x1", "t", "mask"))
    losses, _ = tr.forward_backward(x0, x1, mask, t)
    assert relerr(losses["total"], GM["loss/total"]) < TOL and relerr(losses["flow"], GM["loss/flow"]) < TOL
    got = {}
    for pre, m in (("encoder.", enc), ("flow_decoder.", fdec), ("seg_decoder.", sdec)):
        got.update({pre + k: p.grad.clone() for k, p in m.named_parameters()})
    check_grads(got, sub(GM, "grad/"), TOL)
