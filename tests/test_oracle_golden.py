"""Pins oracle/unet_oracle.py against vectors produced by the reference's own modules
(tests/golden/make_golden.py).  CPU only.  Tolerance: 1e-5 relative (max-norm) for
forward values, 2e-4 for gradients / post-Adam state (same fp32 ATen conv kernels on
both sides; the slack covers summation-order differences of the hand-written BN,
pooling and bilinear restatements and autograd's accumulation order)."""
import pytest
import torch

from conftest import relerr, sub
from oracle import unet_oracle as O

FWD_TOL = 1e-5
GRAD_TOL = 2e-4


def _grad_ok(got, ref, scale):
    # bias gradients in front of a train-mode BatchNorm are analytically zero: compare
    # against the layer's weight-gradient scale instead of their own (noise) magnitude.
    return float((got - ref).abs().max()) <= GRAD_TOL * max(float(ref.abs().max()), scale)


@pytest.mark.parametrize("name", ["tiny", "odd3"])
def test_train_steps_match_reference(name, golden_tiny, golden_odd3):
    G = golden_tiny if name == "tiny" else golden_odd3
    P = sub(G, "init/")
    steps = 0
    while f"step{steps}/x0" in G:
        steps += 1
    batches = [(G[f"step{s}/x0"], G[f"step{s}/x1"], G[f"step{s}/t"]) for s in range(steps)]
    # the full multi-step run (Adam moments carried across steps): EVERY step's velocity, loss and gradients against the
    # reference's -- step 1 runs on the oracle's own post-Adam parameters, so its comparison also covers the update
    P_fin, hist = O.train_steps(P, batches, lr=1e-4, weight_decay=1e-5)
    assert len(hist) == steps and (steps >= 2 or name != "tiny")      # (odd3 holds one step)
    for s in range(steps):
        # (later steps inherit the ~1e-6 parameter differences of the steps before them)
        ftol, gmul = (FWD_TOL, 1.0) if s == 0 else (20 * FWD_TOL, 5.0)
        assert relerr(hist[s]["v"], G[f"step{s}/v"]) < ftol, s
        assert relerr(hist[s]["loss"], G[f"step{s}/loss"]) < ftol, s
        gref = sub(G, f"step{s}/grad/")
        gscale = max(float(v.abs().max()) for v in gref.values())
        for k, g in hist[s]["grads"].items():
            assert float((g - gref[k]).abs().max()) <= gmul * GRAD_TOL * max(float(gref[k].abs().max()), 1e-3 * gscale), (s, k)
    last = sub(G, f"step{steps - 1}/after/")
    for k, v in last.items():
        if k.endswith("num_batches_tracked"):
            assert int(P_fin[k]) == int(v), k
        else:
            # Adam's update is ~lr*sign(g) on its first steps, so a parameter that starts at 0
            # (BN beta) carries lr-sized values whose relative error mirrors the gradient's
            # noise; allow 2% of one lr-step absolute on top of the relative bound.
            err = float((P_fin[k] - v).abs().max())
            slack = 0.02
            if k.endswith(("double_conv.0.bias", "double_conv.3.bias")):
                # conv bias in front of a train-mode BatchNorm: its true gradient is exactly 0,
                # the reference's is rounding noise, and Adam rescales that noise to O(lr)
                # steps -- the reference value itself is not reproducible beyond |dp| <= lr/step.
                slack = 1.0
            assert err <= GRAD_TOL * float(v.abs().max()) + slack * 1e-4 * steps, k
    assert relerr(hist[-1]["loss"], G[f"step{steps - 1}/loss"]) < FWD_TOL


def test_eval_forward_and_euler(golden_tiny):
    G = golden_tiny
    P = sub(G, "step1/after/")
    v = O.flow_forward(G["eval/t"], G["step1/x0"][:2], P, False)
    assert relerr(v, G["eval/v"]) < FWD_TOL
    x = O.euler_sample(P, G["euler/x_start"], int(G["euler/n_steps"]))
    assert relerr(x, G["euler/x_end"]) < 1e-4


def test_time_embedding(golden_ops):
    for dim in (32, 256):
        y = O.time_embedding(golden_ops[f"temb{dim}/t"], dim)
        assert relerr(y, golden_ops[f"temb{dim}/y"]) < 1e-6


def test_double_conv_fwd_bwd(golden_ops):
    G = golden_ops
    P = {"dc." + k: v.clone() for k, v in sub(G, "dc/init/").items()}
    keys = [k for k in P if k.endswith(("weight", "bias"))]
    for k in keys:
        P[k].requires_grad_(True)
    x = G["dc/x"].clone().requires_grad_(True)
    nb = {}
    y = O.double_conv(x, P, "dc", True, nb)
    assert relerr(y, G["dc/y"]) < FWD_TOL
    gs = torch.autograd.grad((y * G["dc/w"]).sum(), [x] + [P[k] for k in keys])
    assert relerr(gs[0], G["dc/dx"]) < GRAD_TOL
    gscale = max(float(G["dc/grad/" + k[3:]].abs().max()) for k in keys)
    for k, g in zip(keys, gs[1:]):
        assert _grad_ok(g, G["dc/grad/" + k[3:]], 1e-3 * gscale), k
    for k, v in nb.items():
        ref = G["dc/after/" + k[3:]]
        assert (int(v) == int(ref)) if k.endswith("tracked") else relerr(v, ref) < FWD_TOL, k
    Pe = {"dc." + k: v for k, v in sub(G, "dc/after/").items()}
    assert relerr(O.double_conv(G["dc/x"], Pe, "dc", False), G["dc/y_eval"]) < FWD_TOL


def test_down_block(golden_ops):
    G = golden_ops
    P = {"d.maxpool_conv.1." + k[len("maxpool_conv.1."):]: v.clone() for k, v in sub(G, "down/init/").items()}
    x = G["down/x"].clone().requires_grad_(True)
    y = O.double_conv(O.maxpool2(x), P, "d.maxpool_conv.1", True, {})
    assert relerr(y, G["down/y"]) < FWD_TOL
    (dx,) = torch.autograd.grad((y * G["down/w"]).sum(), [x])
    assert relerr(dx, G["down/dx"]) < GRAD_TOL


def test_up_block_with_pad(golden_ops):
    G = golden_ops
    P = {"u." + k: v.clone() for k, v in sub(G, "up/init/").items()}
    lo = G["up/lo"].clone().requires_grad_(True)
    sk = G["up/skip"].clone().requires_grad_(True)
    y = O.up_block(lo, sk, P, "u", True, {})
    assert relerr(y, G["up/y"]) < FWD_TOL
    dlo, dsk = torch.autograd.grad((y * G["up/w"]).sum(), [lo, sk])
    assert relerr(dlo, G["up/dlo"]) < GRAD_TOL
    assert relerr(dsk, G["up/dskip"]) < GRAD_TOL


def test_bilinear_and_pool(golden_ops):
    assert relerr(O.upsample2x_bilinear_ac(golden_ops["bilinear/x"]), golden_ops["bilinear/y"]) < 1e-6
    assert torch.equal(O.maxpool2(golden_ops["pool/x"]), golden_ops["pool/y"])


def test_multitask_step_matches_reference():
    """Row f2: shared encoder run twice, flow head + mask head, Dice + BCE."""
    from conftest import load_golden
    G = load_golden("multitask_step.npz")
    P = sub(G, "init/")
    losses, grads, nb = O.multitask_loss_and_grads(P, G["x0"], G["x1"], G["t"], G["mask"], 1.0, 0.5)
    for k in ("total", "flow", "dice", "bce"):
        assert relerr(losses[k], G["loss/" + k]) < FWD_TOL, k
    gref = sub(G, "grad/")
    gscale = max(float(v.abs().max()) for v in gref.values())
    for k, g in grads.items():
        assert _grad_ok(g, gref[k], 1e-3 * gscale), k
    for k, v in nb.items():
        ref = G["after/" + k]
        assert (int(v) == int(ref)) if k.endswith("tracked") else relerr(v, ref) < FWD_TOL, k
    assert int(nb["encoder.inc.double_conv.1.num_batches_tracked"]) == 2      # the encoder ran twice


def test_multiclass_step_matches_reference():
    """Row f2, multiclass form: 5-class mask head, softmax Dice + cross entropy."""
    from conftest import load_golden
    G = load_golden("multiclass_step.npz")
    P = sub(G, "init/")
    losses, grads, _ = O.multitask_loss_and_grads(P, G["x0"], G["x1"], G["t"], G["mask"], 1.0, 0.5, multiclass=True)
    for k in ("total", "flow", "dice", "ce"):
        assert relerr(losses[k], G["loss/" + k]) < FWD_TOL, k
    gref = sub(G, "grad/")
    gscale = max(float(v.abs().max()) for v in gref.values())
    for k, g in grads.items():
        assert _grad_ok(g, gref[k], 1e-3 * gscale), k
    z = G["lossop/z"].clone().requires_grad_(True)
    seg, d, ce = O.seg_loss_multiclass(z, G["lossop/target"], 0.3, ignore_index=2)
    seg.backward()
    assert relerr(d, G["lossop/dice"]) < 1e-6 and relerr(ce, G["lossop/ce"]) < 1e-6
    assert relerr(z.grad, G["lossop/dz"]) < 1e-5


def test_loss_and_conditioning_variants_match_reference():
    """Row f4: mask as 4th input channel, ROI-weighted MSE, Charbonnier ROI value."""
    from conftest import load_golden
    G = load_golden("variants_step.npz")
    P = sub(G, "init/")
    for tag, lam in (("mse", None), ("roi", 10.0)):
        loss, v, grads, _ = O.variant_loss_and_grads(P, G["x0"], G["x1"], G["t"], G["mask"], mask_as_channel=True,
                                                     roi_lambda=lam)
        assert relerr(loss, G[tag + "/loss"]) < FWD_TOL and relerr(v, G[tag + "/v"]) < FWD_TOL
        gref = sub(G, tag + "/grad/")
        gscale = max(float(x.abs().max()) for x in gref.values())
        for k, g in grads.items():
            assert _grad_ok(g, gref[k], 1e-3 * gscale), (tag, k)
    xt, _ = O.cfm_sample(G["x0"], G["x1"], G["t"])
    assert relerr(O.charbonnier_roi(xt, G["x1"], G["mask"]), G["charb/value"]) < 1e-6
