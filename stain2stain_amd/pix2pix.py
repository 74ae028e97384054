"""Building blocks of the pix2pix path named by BASELINE.json's north_star.

The reference repository contains no pix2pix model, so nothing here mirrors a reference class: the
module below has the signature of ``nn.Sequential(nn.InstanceNorm2d(C, eps, affine), nn.LeakyReLU(slope))`` - the
norm + activation pair of the pix2pix U-Net generator's encoder and of the PatchGAN discriminator (slope 0.2), and
with ``negative_slope=0`` the decoder's InstanceNorm + ReLU - and is checked against exactly that torch pair.
State-dict keys follow nn.InstanceNorm2d (``weight`` / ``bias`` when affine, no running statistics).

Activations are NHWC on the device: a logical [B,C,H,W] tensor in torch.channels_last memory format is used in place
(zero copies); any other layout is converted once on the way in.  HIP-only, like the rest of the package.
"""
from __future__ import annotations

import torch
from torch import nn

from . import ops


class _InstNormLReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, slope):
        xs = x.permute(0, 2, 3, 1)                                   # NHWC view of a channels_last tensor
        y, stats = ops.instnorm_lrelu_fwd(xs, weight, bias, eps, slope)
        ctx.save_for_backward(xs, stats)
        ctx.slope, ctx.affine = slope, weight is not None
        return y.permute(0, 3, 1, 2)                                 # logical NCHW, channels_last memory

    @staticmethod
    def backward(ctx, gy):
        xs, stats = ctx.saved_tensors
        g = gy.permute(0, 2, 3, 1)
        if not g.is_contiguous():
            g = g.contiguous()
        g = g.to(xs.dtype)
        dg = db = None
        if ctx.affine:
            C = xs.shape[3]
            dg = torch.empty(C, dtype=torch.float32, device=xs.device)
            db = torch.empty(C, dtype=torch.float32, device=xs.device)
        dx = ops.instnorm_lrelu_bwd(g, xs, stats, ctx.slope, dg, db)
        return dx.permute(0, 3, 1, 2), dg, db, None, None


class InstanceNormLeakyReLU(nn.Module):
    """``InstanceNorm2d(num_features, eps, affine) -> LeakyReLU(negative_slope)`` in one forward and one backward
    pass pair over HBM.  ``precision``: "bf16" stores activations in bf16 (statistics in fp32), "fp32" keeps fp32."""

    def __init__(self, num_features: int, eps: float = 1e-5, affine: bool = False, negative_slope: float = 0.2,
                 precision: str = "bf16"):
        super().__init__()
        if num_features % 8:
            raise ValueError("InstanceNormLeakyReLU: num_features must be a multiple of 8")
        if precision not in ("bf16", "fp32"):
            raise ValueError("precision must be 'bf16' or 'fp32'")
        self.num_features, self.eps, self.negative_slope = num_features, eps, negative_slope
        self.dtype = torch.bfloat16 if precision == "bf16" else torch.float32
        if affine:
            self.weight = nn.Parameter(torch.ones(num_features))
            self.bias = nn.Parameter(torch.zeros(num_features))
        else:
            self.register_parameter("weight", None)
            self.register_parameter("bias", None)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("stain2stain_amd: InstanceNormLeakyReLU runs on the GPU only (no CPU fallback)")
        if x.dim() != 4 or x.shape[1] != self.num_features:
            raise ValueError(f"expected [B, {self.num_features}, H, W], got {tuple(x.shape)}")
        if x.shape[2] * x.shape[3] <= 1:
            raise ValueError("Expected more than 1 spatial element when training")          # torch's own message
        x = x.to(self.dtype).contiguous(memory_format=torch.channels_last)
        return _InstNormLReLU.apply(x, self.weight, self.bias, self.eps, self.negative_slope)
