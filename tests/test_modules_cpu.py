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
