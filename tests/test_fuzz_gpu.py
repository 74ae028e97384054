"""Seeded random-shape sweep of the bf16 kernels against fp32 torch on the CPU (same tolerances as
tests/test_ops_gpu.py): ragged sizes, channel tails, two-source inputs, every tile configuration the dispatcher can
pick.  One process, small tensors: a few seconds."""
import random

import pytest
import torch
import torch.nn.functional as F

from conftest import relerr
from test_ops_gpu import nchw, nhwc, rnd

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


def _conv_shapes(n, seed):
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        B = rng.choice([1, 2, 3, 5])
        H, W = rng.randint(3, 70), rng.randint(3, 90)
        c0 = 8 * rng.randint(1, 12)
        c1 = 8 * rng.choice([0, 0, 1, 3, 6])
        cout = 8 * rng.randint(1, 30)
        out.append((B, H, W, c0, c1, cout))
    return out


@pytest.mark.parametrize("case", _conv_shapes(14, 2024))
def test_conv_forward_dgrad_wgrad_random_shapes(case):
    from stain2stain_amd import ops
    B, H, W, c0, c1, cout = case
    cin = c0 + c1
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x = torch.rand(B, cin, H, W, generator=g) * 2 - 1
    w = (torch.rand(cout, cin, 3, 3, generator=g) * 2 - 1) * 0.1
    b = torch.rand(cout, generator=g) - 0.5
    dy = torch.rand(B, cout, H, W, generator=g) * 2 - 1
    xr = rnd(x, BF).requires_grad_(True)
    wr = rnd(w, BF).requires_grad_(True)
    ref = F.conv2d(xr, wr, b, padding=1)
    (ref * rnd(dy, BF)).sum().backward()
    xs = nhwc(x, BF)
    x0, x1 = xs[..., :c0], (xs[..., c0:] if c1 else None)
    wf, wd = ops.pack_conv3x3(w.to(DEV), BF)
    y, stat = ops.conv3x3(x0, x1, wf, b.to(DEV), cout, want_stats=True)
    assert relerr(nchw(y), ref.detach()) < 4e-3
    yb = nchw(y)                                                    # statistics are taken from the stored values
    s = stat.sum(-1).cpu()                 # [2][C][producer workgroups]
    assert float((s[0] - yb.sum((0, 2, 3))).abs().max()) < 1e-4 * float(yb.abs().sum((0, 2, 3)).max())
    assert relerr(s[1], (yb * yb).sum((0, 2, 3))) < 1e-4
    dx, _ = ops.conv3x3(nhwc(dy, BF), None, wd, None, cin)
    assert relerr(nchw(dx), xr.grad) < 4e-3
    gw = torch.empty(cout, cin, 3, 3, device=DEV)
    ops.conv3x3_wgrad(nhwc(dy, BF), x0, x1, gw)
    assert relerr(gw.cpu(), wr.grad) < 1e-4


@pytest.mark.parametrize("seed", range(8))
def test_bn_relu_pool_backward_random_shapes(seed):
    from oracle import unet_oracle as O
    from stain2stain_amd import ops
    rng = random.Random(100 + seed)
    B, H, W, C = rng.choice([1, 2, 4]), rng.randint(2, 41), rng.randint(2, 41), 8 * rng.randint(1, 20)
    pool = rng.random() < 0.6 and H >= 2 and W >= 2
    g = torch.Generator().manual_seed(seed)
    z = rnd(torch.randn(B, C, H, W, generator=g) * 2, BF).requires_grad_(True)
    gamma = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    beta = (torch.rand(C, generator=g) - 0.5).requires_grad_(True)
    y, mean, var = O.batchnorm_train(z, gamma, beta)
    a = y.clamp_min(0).to(BF).float()                              # the stored activation
    a_ref = y.clamp_min(0)
    g1 = rnd(torch.randn(B, C, H, W, generator=g), BF)
    loss = (a_ref * g1).sum()
    gp = None
    if pool:
        p = O.maxpool2(a_ref.detach().to(BF).float() + (a_ref - a_ref.detach()))   # pool decisions on the stored values
        gp = rnd(torch.randn(p.shape, generator=g), BF)
        loss = loss + (p * gp).sum()
    loss.backward()
    zs = nhwc(z.detach(), BF)
    count = B * H * W
    stat = torch.stack([z.detach().sum((0, 2, 3)), (z.detach() ** 2).sum((0, 2, 3))])[..., None].to(DEV)
    st = ops.bn_finalize(stat.contiguous(), count, gamma.detach().to(DEV), beta.detach().to(DEV), None, None, None)
    act, pooled = ops.bn_relu_apply(zs, st[2], st[3], want_pool=pool)
    assert relerr(nchw(act), a) < 4e-3
    dgam = torch.empty(C, device=DEV); dbet = torch.empty(C, device=DEV)
    dz = ops.bn_relu_bwd(nhwc(g1, BF), nhwc(gp, BF) if pool else None, zs, st, gamma.detach().to(DEV), dgam, dbet, None)
    # A random draw has a few values on a decision edge (ReLU at 0, two pool candidates equal after bf16 rounding)
    # where the kernel's fmaf and the oracle's (x-mean)*k+b fall on different sides: that moves a whole gradient
    # entry, so the check is on the fraction of entries that agree and on the overall L2 error, not the max-norm.
    scale = float(z.grad.abs().max())
    err = (nchw(dz) - z.grad).abs()
    assert float((err > 1.5e-2 * scale).float().mean()) < 2e-3
    assert float(err.pow(2).sum().sqrt() / z.grad.pow(2).sum().sqrt()) < 5e-2
    assert relerr(dbet.cpu(), beta.grad) < 3e-2 and relerr(dgam.cpu(), gamma.grad) < 3e-2


@pytest.mark.parametrize("seed", range(6))
def test_upsample_random_shapes(seed):
    from stain2stain_amd import ops
    rng = random.Random(300 + seed)
    B, Hin, Win, C = rng.choice([1, 2, 3]), rng.randint(1, 23), rng.randint(1, 23), 8 * rng.randint(1, 24)
    padH, padW = rng.choice([0, 0, 1]), rng.choice([0, 0, 1])       # the F.pad branch of the reference's Up block
    Hout, Wout = 2 * Hin + padH, 2 * Win + padW
    g = torch.Generator().manual_seed(seed)
    x = rnd(torch.randn(B, C, Hin, Win, generator=g), BF).requires_grad_(True)
    up = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    ref = F.pad(up, [padW // 2, padW - padW // 2, padH // 2, padH - padH // 2])
    dy = rnd(torch.randn(ref.shape, generator=g), BF)
    (ref * dy).sum().backward()
    out = torch.empty(B, Hout, Wout, C, device=DEV, dtype=BF)
    ops.upsample2x_fwd(nhwc(x.detach(), BF), out)
    assert relerr(nchw(out), ref.detach()) < 4e-3
    dx = ops.upsample2x_bwd(nhwc(dy, BF), Hin, Win)
    assert relerr(nchw(dx), x.grad) < 6e-3
