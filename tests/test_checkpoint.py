"""Row f3: checkpoint interchange.  tests/golden/tiny_lightning.ckpt was written from the reference's modules
(make_golden.py: make_checkpoint_fixture); no reference-trained checkpoint exists offline, so this pins the
format (keys, shapes, optimiser-state layout) and the numbers the reference computes after loading it."""
import os

import pytest
import torch

from conftest import GOLDEN, load_golden, relerr, sub

CKPT = os.path.join(GOLDEN, "tiny_lightning.ckpt")


def tiny_net(precision="fp32"):
    from stain2stain_amd import FlowUNet
    return FlowUNet(3, [16, 32], 3, 32, precision=precision)


def test_checkpoint_keys_and_shapes_fit_our_modules(tmp_path):
    from stain2stain_amd import checkpoint as C
    ck = C.read_checkpoint(CKPT)
    assert {"state_dict", "optimizer_states", "epoch", "global_step"} <= set(ck)
    sd = C.extract_state_dict(ck, strip_prefix="net.")
    net = tiny_net()
    own = net.state_dict()
    assert list(sd.keys()) == list(own.keys())                      # same names in the same order
    assert all(tuple(sd[k].shape) == tuple(own[k].shape) and sd[k].dtype == own[k].dtype for k in sd)
    res = C.load_weights(net, CKPT, strip_prefix="net.")
    assert not res.missing_keys and not res.unexpected_keys
    assert all(torch.equal(net.state_dict()[k], sd[k]) for k in sd)
    # torch.compile'd LitModules save "net._orig_mod.encoder..." -- same weights
    wrapped = {"state_dict": {k.replace("net.", "net._orig_mod.", 1): v for k, v in ck["state_dict"].items()}}
    assert list(C.extract_state_dict(wrapped, "net.").keys()) == list(own.keys())
    # and back: what we write is what the reference's loader reads (checkpoint['state_dict'], plain tensors)
    out = tmp_path / "ours.ckpt"
    C.save_checkpoint(str(out), net, ck["optimizer_states"][0], epoch=ck["epoch"], global_step=ck["global_step"])
    back = torch.load(str(out), weights_only=True)
    assert list(back["state_dict"].keys()) == list(own.keys())
    assert all(torch.equal(back["state_dict"][k], sd[k]) for k in sd)
    assert back["optimizer_states"][0]["param_groups"][0]["lr"] == 1e-4


def test_checkpoint_mismatch_is_reported():
    from stain2stain_amd import FlowUNet
    from stain2stain_amd import checkpoint as C
    with pytest.raises(RuntimeError, match="does not fit"):
        C.load_weights(FlowUNet(3, [16, 64], 3, 32), CKPT, strip_prefix="net.")
    with pytest.raises(RuntimeError):                                # missing keys under strict loading
        C.load_weights(tiny_net(), {"state_dict": {}}, strict=True)
    with pytest.raises(TypeError):
        C.extract_state_dict({"state_dict": {"a": 1}})


def test_oracle_eval_on_loaded_checkpoint_matches_reference():
    from oracle import unet_oracle as O
    from stain2stain_amd import checkpoint as C
    E = load_golden("checkpoint_expect.npz")
    P = C.extract_state_dict(C.read_checkpoint(CKPT), "net.")
    v = O.flow_forward(E["eval/t"], E["eval/x"], P, False)
    assert relerr(v, E["eval/v"]) < 1e-5


@pytest.mark.gpu
def test_loaded_checkpoint_eval_matches_reference_gpu():
    from stain2stain_amd import checkpoint as C
    E = load_golden("checkpoint_expect.npz")
    for prec, tol in (("fp32", 1e-3), ("bf16", 3e-2)):
        net = tiny_net(prec)
        C.load_weights(net, CKPT, strip_prefix="net.")
        net = net.cuda().eval()
        with torch.no_grad():
            v = net(E["eval/t"].cuda(), E["eval/x"].cuda())
        assert relerr(v, E["eval/v"]) < tol, prec


@pytest.mark.gpu
def test_resume_training_from_reference_checkpoint_gpu():
    """Weights + torch.optim.Adam moments from the checkpoint, third step on the fused trainer == the reference's."""
    from stain2stain_amd import CFMTrainer
    from stain2stain_amd import checkpoint as C
    E = load_golden("checkpoint_expect.npz")
    ck = C.read_checkpoint(CKPT)
    net = tiny_net("fp32")
    C.load_weights(net, ck, strip_prefix="net.")
    tr = CFMTrainer(net.cuda().train(), lr=1.0, weight_decay=0.5)        # overwritten by the loaded param group
    tr.load_optimizer_state_dict(ck["optimizer_states"][0])
    assert tr.step_count == 2 and tr.lr == 1e-4 and tr.wd == 1e-5
    loss = tr.step(E["resume/x0"].cuda(), E["resume/x1"].cuda(), E["resume/t"].cuda())
    assert relerr(loss, E["resume/loss"]) < 1e-3
    before = C.extract_state_dict(ck, "net.")
    for k, ref in sub(E, "resume/after/").items():
        got = net.state_dict()[k].float().cpu()
        if k.endswith("num_batches_tracked"):
            assert int(got) == int(ref) == 3
        elif "running" in k:
            assert relerr(got, ref) < 1e-3, k
        else:
            # the third Adam update itself (about lr = 1e-4 per element) must match, not just the parameter
            upd_ref, upd = ref - before[k], got - before[k]
            slack = 1.0 if k.endswith(("double_conv.0.bias", "double_conv.3.bias")) else 0.05
            assert float((upd - upd_ref).abs().max()) <= slack * 1e-4 + 1e-6, k
    # and the other direction: our optimiser state loads into torch.optim.Adam as is
    sd = tr.optimizer_state_dict()
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, weight_decay=1e-5)
    opt.load_state_dict(sd)
    assert int(opt.state_dict()["state"][0]["step"]) == 3
    assert len(opt.state_dict()["state"]) == len(list(net.parameters()))
