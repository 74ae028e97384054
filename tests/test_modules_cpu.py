"""Host-side surface of the drop-in modules (no GPU): state_dict keys and shapes against the fixtures written
from the reference's classes, the Lightning-facing methods, and the refusal to run without the GPU."""
import functools

import pytest
import torch

from conftest import load_golden, sub


def test_state_dict_keys_and_shapes_match_the_reference_classes():
    from stain2stain_amd import FlowMatchingDecoder, SegmentationDecoder, SharedEncoder
    G = load_golden("multiclass_step.npz")
    for prefix, mod in (("encoder.", SharedEncoder(3, [16, 32])),
                        ("flow_decoder.", FlowMatchingDecoder(32, [16], 3, 32)),
                        ("seg_decoder.", SegmentationDecoder(32, [16], 5))):
        ref = sub(G, "init/" + prefix)
        sd = mod.state_dict()
        assert list(sd.keys()) == list(ref.keys()), prefix
        assert all(tuple(sd[k].shape) == tuple(ref[k].shape) and sd[k].dtype == ref[k].dtype for k in ref), prefix
    G4 = load_golden("variants_step.npz")
    enc4 = SharedEncoder(4, [16, 32])                       # RGB + mask channel
    assert tuple(enc4.state_dict()["inc.double_conv.0.weight"].shape) == \
        tuple(G4["init/encoder.inc.double_conv.0.weight"].shape) == (16, 4, 3, 3)


def test_constructor_limits_are_reported():
    from stain2stain_amd import FlowMatchingDecoder, SegmentationDecoder, SharedEncoder
    with pytest.raises(ValueError):
        SharedEncoder(7, [16, 32])                          # stem kernels: at most 6 input channels
    with pytest.raises(ValueError):
        SharedEncoder(3, [12, 32])                          # channel widths are multiples of 8
    with pytest.raises(ValueError):
        FlowMatchingDecoder(32, [16], 5, 32)                # fused head: at most 4 outputs
    with pytest.raises(ValueError):
        SegmentationDecoder(32, [16], 9)
    with pytest.raises(NotImplementedError):
        FlowMatchingDecoder(32, [16], 3, 32, bilinear=False)


def test_lightning_surface_of_the_step_modules():
    from stain2stain_amd import (ClassConditionalFlowMatchingModule, ClassConditionalFlowUNet,
                                 ConditionalFlowMatchingModule, FlowMatchingDecoder, FlowUNet,
                                 MaskConditionedFlowMatchingModule, MultiTaskFlowMatchingModule,
                                 ROICharbonnierFlowMatchingModule, ROIWeightedFlowMatchingModule,
                                 SegmentationDecoder, SharedEncoder)
    opt = functools.partial(torch.optim.Adam, lr=1e-4, weight_decay=1e-5)
    sched = functools.partial(torch.optim.lr_scheduler.ReduceLROnPlateau, mode="min", factor=0.1, patience=10)
    net = FlowUNet(3, [16, 32], 3, 32)
    for cls in (ConditionalFlowMatchingModule, ROIWeightedFlowMatchingModule, ROICharbonnierFlowMatchingModule,
                MaskConditionedFlowMatchingModule):
        m = cls(net, optimizer=opt, scheduler=sched)
        for name in ("model_step", "training_step", "validation_step", "test_step", "configure_optimizers", "generate"):
            assert callable(getattr(m, name)), (cls.__name__, name)
        cfg = m.configure_optimizers()
        assert isinstance(cfg["optimizer"], torch.optim.Adam)
        assert cfg["lr_scheduler"]["monitor"] == "val/loss" and cfg["lr_scheduler"]["interval"] == "epoch"
    assert ROIWeightedFlowMatchingModule.roi_lambda == 10.0 and ROICharbonnierFlowMatchingModule.lambda_roi == 1.0
    cc = ClassConditionalFlowMatchingModule(ClassConditionalFlowUNet(3, [16, 32], 3, 32, num_classes=3), optimizer=opt)
    assert "label_emb.weight" in dict(cc.net.named_parameters()) and "optimizer" in cc.configure_optimizers()
    mt = MultiTaskFlowMatchingModule(SharedEncoder(3, [16, 32]), FlowMatchingDecoder(32, [16], 3, 32),
                                     SegmentationDecoder(32, [16], 1), optimizer=opt, time_emb_dim=32)
    n_opt = sum(p.numel() for g in mt.configure_optimizers()["optimizer"].param_groups for p in g["params"])
    n_mod = sum(p.numel() for m in (mt.encoder, mt.flow_decoder, mt.seg_decoder) for p in m.parameters())
    assert n_opt == n_mod
    for name in ("forward_flow", "forward_segmentation", "compute_segmentation_loss", "model_step", "generate"):
        assert callable(getattr(mt, name))


def test_product_path_refuses_host_tensors():
    """No CPU fallback: a forward on host tensors raises instead of silently computing somewhere else."""
    from stain2stain_amd import FlowUNet
    net = FlowUNet(3, [16, 32], 3, 32)
    with pytest.raises(RuntimeError):
        net(torch.rand(2), torch.rand(2, 3, 16, 16))


def test_generate_follows_the_reference_mode_and_solver_rules(monkeypatch):
    """Host logic of generate() (reference conditional_flow_matching.py:133-170, ..._multitask.py:419-484) with the two
    HIP ops of the integrators replaced by their torch one-liners (no GPU here): no solver -> ValueError; the module is
    put in eval mode and left there; the solver's method and tolerances are read from attributes or from the keywords
    of a functools.partial (what Hydra's ``_partial_: true`` builds); sampling helpers restore the mode they found,
    also when the network raises."""
    from stain2stain_amd import (ConditionalFlowMatchingModule, MultiTaskFlowMatchingModule, SolverConfig, euler_generate,
                                 flow_matching, ops)
    monkeypatch.setattr(ops, "axpy_", lambda x, y, a: x.add_(y, alpha=a))
    monkeypatch.setattr(ops, "ode_error_norm", lambda e, a, b, atol, rtol: torch.sqrt(torch.mean(
        (e / (atol + rtol * torch.maximum(a.abs(), b.abs()))) ** 2)).reshape(1))

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.bn = torch.nn.BatchNorm2d(3)
            self.calls = 0

        def forward(self, t, x, **kw):
            self.calls += 1
            assert not self.training and t.shape == (x.shape[0],)
            return -x

    x = torch.rand(2, 3, 4, 4)
    m = ConditionalFlowMatchingModule(Net())
    with pytest.raises(ValueError, match="Solver is not initialized"):
        m.generate(x)
    m = ConditionalFlowMatchingModule(Net(), solver=SolverConfig()).train()
    out = m.generate(x)                                              # dopri5: x(1) = x(0) / e
    assert not m.training and not m.net.training and not m.net.bn.training
    assert float((out - x * torch.exp(torch.tensor(-1.0))).abs().max()) < 1e-3
    n_adaptive = m.net.calls
    m.net.calls = 0
    out = m.generate(x, num_steps=10, method="euler")
    assert m.net.calls == 10 and float((out - x * 0.9 ** 10).abs().max()) < 1e-6
    # a partial (Hydra) carries its settings in .keywords; attribute-less objects fall back to dopri5 / 1e-4
    part = functools.partial(SolverConfig, solver="euler")
    assert flow_matching._solver_setting(part, "solver", "dopri5") == "euler"
    assert flow_matching._solver_setting(object(), "atol", 1e-4) == 1e-4
    m2 = ConditionalFlowMatchingModule(Net(), solver=part)
    m2.generate(x, num_steps=3)
    assert m2.net.calls == 3 and n_adaptive > 6
    # helper functions: mode restored, also on failure
    net = Net().train()
    euler_generate(net, x, 2)
    assert net.training

    class Boom(Net):
        def forward(self, t, x):
            raise RuntimeError("boom")

    b = Boom().train()
    with pytest.raises(RuntimeError, match="boom"):
        euler_generate(b, x, 2)
    assert b.training
    # multitask: generate() must not leave the module in train mode -- it stays in eval like the reference
    class Enc(torch.nn.Module):
        def forward(self, x):
            return x, []

    class Dec(torch.nn.Module):
        def forward(self, b, skips, temb=None):
            return -b if temb is not None else b[:, :1]

    mt = MultiTaskFlowMatchingModule(Enc(), Dec(), Dec(), solver=SolverConfig("euler"), time_emb_dim=8)
    monkeypatch.setattr(type(mt.time_embedding), "forward", lambda self, t: t[:, None].expand(-1, 8))
    mt.train()
    img, pm = mt.generate(x, num_steps=4)
    assert not mt.training and img.shape == x.shape and pm.shape == (2, 1, 4, 4)


def test_converted_syncbatchnorm_containers_are_seen_by_the_engine():
    """Lightning's `sync_batchnorm: True` (configs/trainer/ddp.yaml:9) calls torch.nn.SyncBatchNorm.convert_sync_batchnorm,
    which REPLACES the BatchNorm2d modules inside the DoubleConv containers: the engine's layer bundles look their modules
    up on every use (same Parameter objects, same state_dict keys), and without a process group nothing synchronises."""
    import torch
    from stain2stain_amd import FlowUNet, engine
    net = FlowUNet(3, [16, 32], 3, 32)
    keys = list(net.state_dict().keys())
    w = net.encoder._blocks[0][0].bn.weight
    net2 = torch.nn.SyncBatchNorm.convert_sync_batchnorm(net)
    cb = net2.encoder._blocks[0][0]
    assert isinstance(cb.bn, torch.nn.SyncBatchNorm) and cb.bn.weight is w and list(net2.state_dict().keys()) == keys
    assert engine._module_exchange(cb.bn) is None            # torch.distributed not initialised: local statistics
    assert engine._module_exchange(torch.nn.BatchNorm2d(8)) is None
