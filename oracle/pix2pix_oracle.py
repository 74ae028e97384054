"""CPU restatement of the pix2pix building blocks of SURVEY.md section 8, row a13.  TEST INFRASTRUCTURE ONLY: imported
by tests/ (and nothing else); the product path never touches it.

PARITY UNPINNED with respect to the reference repository: it contains no pix2pix / PatchGAN code, no
golden vector and no test for this row.  What pins these formulas instead is torch's own operators, which is what a
pix2pix implementation in the reference's framework would call: tests/test_instnorm_cpu.py checks
``instance_norm_lrelu`` against ``F.leaky_relu(F.instance_norm(...))`` and its gradient against torch's autograd.
"""
from typing import Optional

import torch
from torch import Tensor


def instance_norm_lrelu(x: Tensor, gamma: Optional[Tensor], beta: Optional[Tensor], eps: float = 1e-5,
                        slope: float = 0.2) -> Tensor:
    """nn.InstanceNorm2d(C, eps, affine=gamma is not None, track_running_stats=False) -> nn.LeakyReLU(slope) on an
    NCHW tensor: per-(sample, channel) mean and biased variance over H x W."""
    mean = x.mean(dim=(2, 3), keepdim=True)
    var = ((x - mean) ** 2).mean(dim=(2, 3), keepdim=True)
    z = (x - mean) * torch.rsqrt(var + eps)
    if gamma is not None:
        z = z * gamma[None, :, None, None] + beta[None, :, None, None]
    return torch.where(z > 0, z, slope * z)


def instance_norm_lrelu_grads(x: Tensor, gamma: Optional[Tensor], beta: Optional[Tensor], g: Tensor, eps: float = 1e-5,
                              slope: float = 0.2):
    """Closed-form backward of the above for the cotangent g: returns (dx, dgamma, dbeta)."""
    hw = x.shape[2] * x.shape[3]
    mean = x.mean(dim=(2, 3), keepdim=True)
    var = ((x - mean) ** 2).mean(dim=(2, 3), keepdim=True)
    inv = torch.rsqrt(var + eps)
    xh = (x - mean) * inv
    ga = gamma[None, :, None, None] if gamma is not None else 1.0
    z = xh * ga + (beta[None, :, None, None] if beta is not None else 0.0)
    dz = torch.where(z > 0, g, slope * g)
    c1 = dz.sum(dim=(2, 3), keepdim=True) / hw
    c2 = (dz * xh).sum(dim=(2, 3), keepdim=True) / hw
    dx = ga * inv * (dz - c1 - xh * c2)
    if gamma is None:
        return dx, None, None
    return dx, (dz * xh).sum(dim=(0, 2, 3)), dz.sum(dim=(0, 2, 3))
