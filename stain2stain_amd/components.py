"""Drop-in network components: same constructor signatures, call signatures and ``state_dict`` keys as
the reference's torch-only modules, computed by the HIP kernels.

  reference class (file:line)                                          here
  -------------------------------------------------------------------  -------------------------
  SharedEncoder        src/models/components/shared_encoder.py:40-104   SharedEncoder
  TimeEmbedding        src/models/components/shared_encoder.py:107-135  TimeEmbedding
  FlowMatchingDecoder  src/models/components/task_decoders.py:53-134    FlowMatchingDecoder
  (forward_flow)       src/models/conditional_flow_matching_multitask.py:134-155   FlowUNet (net(t, x))

A Hydra model YAML selects them by changing only the ``_target_`` strings
(``stain2stain_amd.SharedEncoder`` ...), see INTEGRATION.md.  The sub-module tree mirrors the
reference's (``inc.double_conv.0`` ... ``downs.N.maxpool_conv.1.double_conv.4``,
``ups.N.conv.double_conv.*``, ``time_mlp.{0,2}``, ``time_proj``, ``outc``) so checkpoints load by key
and PyTorch's default initialisation consumes the RNG in the same order; the torch.nn leaves are
used as parameter containers only and are never called.

Numerics: ``precision="bf16"`` (default) runs bf16 MFMA with fp32 accumulation, fp32 BatchNorm
statistics and fp32 master weights; ``precision="fp32"`` keeps fp32 activations and forms every MFMA
product from a hi/lo bf16 split (the parity mode checked against the fp32 oracle at 1e-3).

There is no CPU implementation behind these modules: calling them on CPU tensors raises.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import engine, ops

_PRECISIONS = {"bf16": torch.bfloat16, "fp32": torch.float32, "float32": torch.float32,
               "bfloat16": torch.bfloat16}


def _resolve_precision(p) -> torch.dtype:
    if isinstance(p, torch.dtype):
        if p in (torch.bfloat16, torch.float32):
            return p
    elif str(p) in _PRECISIONS:
        return _PRECISIONS[str(p)]
    raise ValueError(f"precision must be 'bf16' or 'fp32', got {p!r}")


def _double_conv_container(cin: int, cout: int) -> nn.Module:
    """Parameter container with the reference DoubleConv's layout: double_conv.{0,1,3,4}."""
    holder = nn.Module()
    holder.double_conv = nn.Sequential(
        nn.Conv2d(cin, cout, kernel_size=3, padding=1), nn.BatchNorm2d(cout), nn.ReLU(inplace=True),
        nn.Conv2d(cout, cout, kernel_size=3, padding=1), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))
    return holder


def _bundles(holder: nn.Module, prefix: str) -> Tuple[engine.ConvBN, engine.ConvBN]:
    seq = holder.double_conv
    p = f"{prefix}.double_conv"
    return engine.ConvBN(p, "0", "1", seq), engine.ConvBN(p, "3", "4", seq)


def _check_channels(chs: Sequence[int]) -> None:
    bad = [c for c in chs if c % 8]
    if bad:
        raise ValueError(f"stain2stain_amd kernels need feature widths that are multiples of 8, got {bad}")


def _as_nhwc(t: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """NCHW-shaped tensor -> NHWC view in the compute dtype (zero-copy when it already is one of ours)."""
    if not t.is_cuda:
        raise RuntimeError("stain2stain_amd: CPU tensors are not supported (HIP-only implementation)")
    v = t.permute(0, 2, 3, 1)
    if t.dtype == dtype and v.stride(3) == 1:
        try:
            ops._nhwc(v)
            return v
        except RuntimeError:
            pass
    if t.dtype == torch.float32 and t.is_contiguous() and t.shape[1] % 8 == 0:
        return ops.nchw_to_nhwc(t, dtype)
    return v.contiguous().to(dtype)   # foreign layout/dtype: plain copy at the boundary


def _fresh_grads(named: Sequence[Tuple[str, torch.Tensor]]) -> Dict[str, torch.Tensor]:
    return {n: torch.empty_like(p, dtype=torch.float32) for n, p in named}


class _side_wgrads:
    """Inside a module's autograd backward: the conv weight gradients go to a second HIP stream (engine.run_on_side), where
    they run beside the bandwidth-bound BatchNorm / up-sampling backward passes as in the fused trainer, and the compute
    stream joins it before the gradients are handed back to autograd."""

    def __init__(self, device):
        self.side = ops.shared_side_stream(device)

    def __enter__(self):
        self.prev, engine.side_stream = engine.side_stream, self.side
        return self

    def __exit__(self, *exc):
        engine.side_stream = self.prev
        engine.join_side(self.side)
        return False


# ------------------------------------------------------------------------------------------------
class TimeEmbedding(nn.Module):
    """Sinusoidal embedding of the raw flow time t in [0, 1] (no parameters)."""

    def __init__(self, dim: int):
        super().__init__()
        self.dim = dim

    def forward(self, t: torch.Tensor) -> torch.Tensor:
        return ops.time_embedding(t.detach(), self.dim)


# ------------------------------------------------------------------------------------------------
class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod: "SharedEncoder", x: torch.Tensor, *params):
        ectx = engine.encoder_forward(mod._blocks, x.detach().contiguous().float(), mod.compute_dtype, mod.training)
        ctx.mod, ctx.ectx = mod, ectx
        ctx.set_materialize_grads(False)
        outs = [f.permute(0, 3, 1, 2) for f in ectx.feats]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dfeats):
        mod, ectx = ctx.mod, ctx.ectx
        if ectx.layers[0][0] is None:
            raise RuntimeError("stain2stain_amd: backward through an eval-mode forward is not supported")
        named = list(mod.named_parameters())
        grads = _fresh_grads(named)
        df = [None if g is None else _as_nhwc(g, ectx.dtype) for g in dfeats]
        with _side_wgrads(ectx.x_nchw.device):
            engine.encoder_backward(mod._blocks, ectx, df, grads)
        ctx.ectx = None
        return (None, None) + tuple(grads[n] for n, _ in named)


class SharedEncoder(nn.Module):
    """U-Net encoder: ``inc`` DoubleConv then N x (MaxPool2d(2) -> DoubleConv).

    forward(x[B,C,H,W]) -> (bottleneck, skips deepest-first), like the reference.  The returned tensors
    are NCHW-shaped views over NHWC storage in the compute dtype.
    """

    def __init__(self, in_channels: int = 3, features: Optional[List[int]] = None,
                 return_skip_connections: bool = True, precision: str = "bf16"):
        super().__init__()
        if features is None:
            features = [64, 128, 256, 512, 1024]
        features = list(features)
        _check_channels(features)
        if in_channels > 6:
            raise ValueError("stain2stain_amd stem kernels support at most 6 input channels (RGB tiles + condition)")
        self.in_channels = in_channels
        self.features = features
        self.return_skip_connections = return_skip_connections
        self.compute_dtype = _resolve_precision(precision)
        self.inc = _double_conv_container(in_channels, features[0])
        self.downs = nn.ModuleList()
        for i in range(len(features) - 1):
            down = nn.Module()
            down.maxpool_conv = nn.Sequential(nn.MaxPool2d(2), _double_conv_container(features[i], features[i + 1]))
            self.downs.append(down)
        self._blocks = [_bundles(self.inc, "inc")] + [
            _bundles(d.maxpool_conv[1], f"downs.{i}.maxpool_conv.1") for i, d in enumerate(self.downs)]

    def forward(self, x: torch.Tensor):
        feats = _EncoderFn.apply(self, x, *self.parameters())
        bottleneck = feats[-1]
        if self.return_skip_connections:
            return bottleneck, list(feats[:-1][::-1])
        return bottleneck, []


# ------------------------------------------------------------------------------------------------
class _DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod: "FlowMatchingDecoder", n_skips: int, bottleneck, t_emb, *rest):
        skips = rest[:n_skips]
        dt = mod.compute_dtype
        dctx = engine.decoder_forward(mod, _as_nhwc(bottleneck.detach(), dt),
                                      [_as_nhwc(s.detach(), dt) for s in skips],
                                      t_emb.detach().float().contiguous(), dt, mod.training)
        ctx.mod, ctx.dctx, ctx.n_skips = mod, dctx, n_skips
        ctx.need_temb = t_emb.requires_grad
        return dctx.v

    @staticmethod
    def backward(ctx, dv):
        mod, dctx = ctx.mod, ctx.dctx
        if dctx.layers and dctx.layers[0][0] is None:
            raise RuntimeError("stain2stain_amd: backward through an eval-mode forward is not supported")
        named = list(mod.named_parameters())
        grads = _fresh_grads(named)
        with _side_wgrads(dv.device):
            dbott, dskips, dtemb = engine.decoder_backward(mod, dctx, dv.contiguous().float(), grads,
                                                           need_dt_emb=ctx.need_temb)
        ctx.dctx = None
        outs = (None, None, dbott.permute(0, 3, 1, 2), dtemb) + tuple(d.permute(0, 3, 1, 2) for d in dskips)
        return outs + tuple(grads[n] for n, _ in named)


class FlowMatchingDecoder(nn.Module):
    """Velocity head: time MLP -> add to bottleneck -> N x Up(bilinear x2, cat skip, DoubleConv) -> Conv1x1."""

    def __init__(self, bottleneck_channels: int = 1024, features: Optional[List[int]] = None, out_channels: int = 3,
                 time_emb_dim: int = 256, bilinear: bool = True, precision: str = "bf16"):
        super().__init__()
        if features is None:
            features = [512, 256, 128, 64]
        features = list(features)
        if not bilinear:
            raise NotImplementedError("bilinear=False (ConvTranspose2d) is a dead branch in the reference configs")
        _check_channels([bottleneck_channels] + features)
        if out_channels > 4:
            raise ValueError("stain2stain_amd head kernel supports at most 4 output channels")
        self.bottleneck_channels = bottleneck_channels
        self.time_emb_dim = time_emb_dim
        self.compute_dtype = _resolve_precision(precision)
        self.time_mlp = nn.Sequential(nn.Linear(time_emb_dim, time_emb_dim), nn.SiLU(),
                                      nn.Linear(time_emb_dim, time_emb_dim))
        self.time_proj = nn.Linear(time_emb_dim, bottleneck_channels)
        self.ups = nn.ModuleList()
        in_ch = bottleneck_channels
        for feat in features:
            up = nn.Module()
            up.up = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
            up.conv = _double_conv_container(in_ch + feat, feat)
            self.ups.append(up)
            in_ch = feat
        self.outc = nn.Conv2d(features[-1], out_channels, kernel_size=1)
        self.up_blocks = [_bundles(u.conv, f"ups.{i}.conv") for i, u in enumerate(self.ups)]

    def forward(self, bottleneck: torch.Tensor, skip_connections: List[torch.Tensor], t_emb: torch.Tensor):
        skips = list(skip_connections)[: len(self.ups)]
        return _DecoderFn.apply(self, len(skips), bottleneck, t_emb, *skips, *self.parameters())


class _SegDecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod: "SegmentationDecoder", n_skips: int, bottleneck, *rest):
        skips = rest[:n_skips]
        dt = mod.compute_dtype
        dctx = engine.decoder_forward(mod, _as_nhwc(bottleneck.detach(), dt), [_as_nhwc(s.detach(), dt) for s in skips],
                                      None, dt, mod.training)
        ctx.mod, ctx.dctx = mod, dctx
        return dctx.v

    @staticmethod
    def backward(ctx, dv):
        mod, dctx = ctx.mod, ctx.dctx
        if dctx.layers and dctx.layers[0][0] is None:
            raise RuntimeError("stain2stain_amd: backward through an eval-mode forward is not supported")
        named = list(mod.named_parameters())
        grads = _fresh_grads(named)
        with _side_wgrads(dv.device):
            dbott, dskips, _ = engine.decoder_backward(mod, dctx, dv.contiguous().float(), grads)
        ctx.dctx = None
        outs = (None, None, dbott.permute(0, 3, 1, 2)) + tuple(d.permute(0, 3, 1, 2) for d in dskips)
        return outs + tuple(grads[n] for n, _ in named)


class SegmentationDecoder(nn.Module):
    """Mask head (reference src/models/components/task_decoders.py:137-194): N x Up(bilinear x2, cat skip,
    DoubleConv) -> Conv1x1, no time conditioning.  forward(bottleneck, skips) -> logits [B, out_channels, H, W]."""

    def __init__(self, bottleneck_channels: int = 1024, features: Optional[List[int]] = None, out_channels: int = 1,
                 bilinear: bool = True, precision: str = "bf16"):
        super().__init__()
        if features is None:
            features = [512, 256, 128, 64]
        features = list(features)
        if not bilinear:
            raise NotImplementedError("bilinear=False (ConvTranspose2d) is a dead branch in the reference configs")
        _check_channels([bottleneck_channels] + features)
        if out_channels > 8:
            raise ValueError("stain2stain_amd segmentation head kernel supports at most 8 output channels")
        self.compute_dtype = _resolve_precision(precision)
        self.time_mlp = None
        self.ups = nn.ModuleList()
        in_ch = bottleneck_channels
        for feat in features:
            up = nn.Module()
            up.up = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
            up.conv = _double_conv_container(in_ch + feat, feat)
            self.ups.append(up)
            in_ch = feat
        self.outc = nn.Conv2d(features[-1], out_channels, kernel_size=1)
        self.up_blocks = [_bundles(u.conv, f"ups.{i}.conv") for i, u in enumerate(self.ups)]

    def forward(self, bottleneck: torch.Tensor, skip_connections: List[torch.Tensor]):
        skips = list(skip_connections)[: len(self.ups)]
        return _SegDecoderFn.apply(self, len(skips), bottleneck, *skips, *self.parameters())


# ------------------------------------------------------------------------------------------------
class FlowUNet(nn.Module):
    """``net(t, x, y=None, **kwargs) -> v``: encoder + time embedding + flow decoder in one module.

    This is the ``net:`` target for ConditionalFlowMatchingLitModule
    (src/models/conditional_flow_matching.py:44-51) built from the in-repo U-Net, i.e. what
    MultiTaskFlowMatchingLitModule.forward_flow composes (conditional_flow_matching_multitask.py:134-155).
    Scalar / 1-element ``t`` is broadcast over the batch as the reference's FlowWrapper does (:457-465).
    """

    def __init__(self, in_channels: int = 3, features: Optional[List[int]] = None, out_channels: int = 3,
                 time_emb_dim: int = 256, precision: str = "bf16"):
        super().__init__()
        if features is None:
            features = [64, 128, 256, 512, 1024]
        self.encoder = SharedEncoder(in_channels, features, True, precision)
        self.flow_decoder = FlowMatchingDecoder(features[-1], list(features[:-1][::-1]), out_channels, time_emb_dim,
                                                True, precision)
        self.time_embedding = TimeEmbedding(time_emb_dim)

    def forward(self, t: torch.Tensor, x: torch.Tensor, y=None, **kwargs) -> torch.Tensor:
        if t.dim() == 0:
            t = t.unsqueeze(0).expand(x.shape[0])
        elif t.dim() == 1 and t.shape[0] == 1:
            t = t.expand(x.shape[0])
        bottleneck, skips = self.encoder(x)
        return self.flow_decoder(bottleneck, skips, self.time_embedding(t))


class _ClassEmbedAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, temb: torch.Tensor, table: torch.Tensor, y: torch.Tensor):
        ctx.save_for_backward(y)
        ctx.num_classes = table.shape[0]
        return ops.class_embed_add(temb.detach().float(), table.detach(), y)

    @staticmethod
    def backward(ctx, dout):
        (y,) = ctx.saved_tensors
        return None, ops.class_embed_bwd(dout.float(), y, ctx.num_classes), None


class ClassConditionalFlowUNet(FlowUNet):
    """``net(t, x, y) -> v`` for ClassConditionalFlowMatchingLitModule (class_conditional_flow_matching.py:39-47,
    configs/model/class_conditional_flow_matching.yaml:19-22 ``num_classes: 3``).  The reference's network there is
    the third-party torchcfm UNetModel (absent); this is the in-repo U-Net with a learned ``label_emb`` row added
    to the sinusoidal time embedding -- build-defined, parity unpinned (SURVEY section 8 d cfg5 / f4).  With
    ``y=None`` it is the unconditional FlowUNet."""

    def __init__(self, in_channels: int = 3, features: Optional[List[int]] = None, out_channels: int = 3,
                 time_emb_dim: int = 256, num_classes: int = 3, precision: str = "bf16"):
        super().__init__(in_channels, features, out_channels, time_emb_dim, precision)
        self.num_classes = num_classes
        self.label_emb = nn.Embedding(num_classes, time_emb_dim)

    def forward(self, t: torch.Tensor, x: torch.Tensor, y: Optional[torch.Tensor] = None, **kwargs) -> torch.Tensor:
        if t.dim() == 0:
            t = t.unsqueeze(0).expand(x.shape[0])
        elif t.dim() == 1 and t.shape[0] == 1:
            t = t.expand(x.shape[0])
        temb = self.time_embedding(t)
        if y is not None:
            y = y.to(torch.int64).reshape(-1)
            if y.shape[0] == 1 and x.shape[0] > 1:
                y = y.expand(x.shape[0])
            y = y.contiguous()
            if y.shape[0] != x.shape[0]:
                raise ValueError(f"y has {y.shape[0]} labels for a batch of {x.shape[0]}")
            if int(y.min()) < 0 or int(y.max()) >= self.num_classes:      # nn.Embedding raises IndexError here
                raise IndexError(f"class label outside [0, {self.num_classes})")
            temb = _ClassEmbedAdd.apply(temb, self.label_emb.weight, y)
        bottleneck, skips = self.encoder(x)
        return self.flow_decoder(bottleneck, skips, temb)
