"""Per-kernel parity on the GPU: every C-ABI entry point against the fp32 CPU oracle ops.

Tolerances (max-norm relative, stated per test):
  * fp32 "split" mode (hi/lo bf16 MFMA, fp32 accumulate) vs the fp32 oracle: 1e-4 (well inside the 1e-3
    north-star bound; expected ~1e-5).
  * bf16 mode vs the oracle evaluated on bf16-rounded operands: 4e-3 for bf16-stored outputs (one bf16
    rounding of the result is 2^-9 = 2e-3), 1e-4 for fp32 outputs (statistics, weight gradients).
"""
import os

import pytest
import torch
import torch.nn.functional as F

from conftest import relerr

pytestmark = pytest.mark.gpu

DEV = "cuda"


def nhwc(x, dtype):
    return x.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)


def nchw(y):
    return y.float().cpu().permute(0, 3, 1, 2).contiguous()


def rnd(x, dtype):
    return x.to(dtype).float() if dtype == torch.bfloat16 else x


def tol_act(dtype):
    return 4e-3 if dtype == torch.bfloat16 else 1e-4


CONV_CASES = [
    # B, H, W, c0, c1, cout
    (2, 20, 37, 16, 0, 24),     # odd sizes, N tail (cout < 64 config)
    (1, 16, 16, 64, 0, 128),    # narrow-image config
    (2, 32, 64, 32, 48, 160),   # two sources (concat), cout tail in the 128 config
    (1, 9, 70, 8, 8, 72),       # chunk straddles the two sources, W tail
    (3, 12, 10, 40, 0, 200),    # narrow + everything ragged
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3x3_forward_and_stats(case, dtype):
    from stain2stain_amd import ops
    B, H, W, c0, c1, cout = case
    g = torch.Generator().manual_seed(7)
    x = torch.rand(B, c0 + c1, H, W, generator=g) * 2 - 1
    w = (torch.rand(cout, c0 + c1, 3, 3, generator=g) * 2 - 1) * 0.1
    b = torch.rand(cout, generator=g) - 0.5
    ref = F.conv2d(rnd(x, dtype), rnd(w, dtype), b, padding=1)
    xs = nhwc(x, dtype)
    # sources as channel slices of one wider buffer (exercises the pixel stride)
    x0 = xs[..., :c0]
    x1 = xs[..., c0:] if c1 else None
    wf, _ = ops.pack_conv3x3(w.to(DEV), dtype)
    y, stat = ops.conv3x3(x0, x1, wf, b.to(DEV), cout, want_stats=True)
    torch.cuda.synchronize()
    assert relerr(nchw(y), ref) < tol_act(dtype)
    s = stat.sum(-1).cpu()                 # [2][C][producer workgroups]
    assert relerr(s[0], ref.sum((0, 2, 3))) < 2e-4 + (1e-3 if dtype == torch.bfloat16 else 0)
    assert relerr(s[1], (ref * ref).sum((0, 2, 3))) < 2e-4 + (1e-3 if dtype == torch.bfloat16 else 0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv3x3_epilogue_affine_relu(dtype):
    from stain2stain_amd import ops
    g = torch.Generator().manual_seed(8)
    x = torch.rand(2, 32, 18, 33, generator=g) * 2 - 1
    w = (torch.rand(64, 32, 3, 3, generator=g) * 2 - 1) * 0.1
    b = torch.rand(64, generator=g) - 0.5
    sc = torch.rand(64, generator=g) + 0.5
    sh = torch.rand(64, generator=g) - 0.5
    ref = (F.conv2d(rnd(x, dtype), rnd(w, dtype), b, padding=1) * sc[None, :, None, None]
           + sh[None, :, None, None]).clamp_min(0)
    wf, _ = ops.pack_conv3x3(w.to(DEV), dtype)
    y, _ = ops.conv3x3(nhwc(x, dtype), None, wf, b.to(DEV), 64, scale=sc.to(DEV), shift=sh.to(DEV), relu=True)
    assert relerr(nchw(y), ref) < tol_act(dtype)


@pytest.mark.parametrize("case", [(1, 16, 16, 512, 256, 192), (2, 8, 8, 256, 0, 64), (1, 32, 32, 320, 0, 256)])
def test_conv3x3_split_k_small_batch(case):
    """Small batches (the reference samples one tile at a time): few output tiles, deep K -- the launch is split over
    the 32-channel chunks and a second launch folds the fp32 partial tiles with bias, folded eval-mode BatchNorm and ReLU.
    Plain (bias only) and fused-epilogue forms, two sources, against torch on bf16-rounded operands."""
    from stain2stain_amd import ops
    B, H, W, c0, c1, cout = case
    dtype = torch.bfloat16
    assert ops._L().s2s_conv3x3_ksplit(0, B, H, W, cout, c0 + c1) > 1
    g = torch.Generator().manual_seed(18)
    x = torch.rand(B, c0 + c1, H, W, generator=g) * 2 - 1
    w = (torch.rand(cout, c0 + c1, 3, 3, generator=g) * 2 - 1) * 0.05
    b = torch.rand(cout, generator=g) - 0.5
    sc, sh = torch.rand(cout, generator=g) + 0.5, torch.rand(cout, generator=g) - 0.5
    conv = F.conv2d(rnd(x, dtype), rnd(w, dtype), b, padding=1)
    xs = nhwc(x, dtype)
    x0, x1 = xs[..., :c0], (xs[..., c0:] if c1 else None)
    wf, _ = ops.pack_conv3x3(w.to(DEV), dtype)
    y, _ = ops.conv3x3(x0, x1, wf, b.to(DEV), cout)
    assert relerr(nchw(y), conv) < tol_act(dtype)
    y, _ = ops.conv3x3(x0, x1, wf, b.to(DEV), cout, scale=sc.to(DEV), shift=sh.to(DEV), relu=True)
    assert relerr(nchw(y), (conv * sc[None, :, None, None] + sh[None, :, None, None]).clamp_min(0)) < tol_act(dtype)
    _, st = ops.conv3x3(x0, x1, wf, b.to(DEV), cout, want_stats=True)            # statistics: never split
    assert st is not None and relerr(st[0].sum(1).cpu(), rnd(conv, dtype).sum((0, 2, 3))) < 1e-3


STAGE_CASES = [
    # B, H, W, c0, c1, cout, stats, level   (>= 512 jobs of 16 x 32 pixels x 64 channels: conv3x3_stage_kernel;
    #  level = the S2S_CONV_STAGE value from which the launch is staged: 1 = default, 2 = also streaming with several
    #  channel tiles -- tests/test_env_variants_gpu.py runs this file under 0 and 2 --, 9 = never)
    (6, 250, 270, 64, 0, 64, True, 1),      # filter resident; ragged bottom / right tiles, XCD-cut order
    (6, 240, 270, 32, 32, 64, True, 1),     # resident, two sources; 810 pixel tiles (not a multiple of 8): plain order
    (2, 250, 270, 64, 0, 192, False, 1),    # resident, three channel tiles: the filter is reloaded twice per workgroup
    (3, 250, 270, 64, 0, 72, False, 1),     # resident, last channel tile holds 8 channels
    (3, 250, 270, 64, 0, 128, True, 1),     # resident with statistics over two channel tiles (flushed at the change)
    (16, 100, 130, 64, 128, 64, True, 1),   # streaming, six chunks from two sources (the decoder's concat)
    (13, 120, 130, 128, 0, 40, False, 1),   # streaming, one partial channel tile (a data gradient's shape)
    (8, 120, 130, 128, 0, 128, True, 2),    # streaming: one channel tile per workgroup, zero rows for the other
    (8, 120, 130, 32, 8, 128, False, 1),    # resident, 40 input channels from two sources: the second chunk holds 8
    (6, 250, 270, 8, 0, 64, True, 1),       # resident, ONE chunk of 8 channels (the stem on its 8-channel image)
    (6, 250, 270, 16, 0, 128, False, 1),    # resident, one chunk, two channel tiles
    (5, 120, 130, 128, 0, 192, False, 2),   # streaming, three channel tiles (a workgroup cycles through them)
    (600, 16, 16, 64, 0, 64, True, 1),      # many small images: every tile is half outside (W = 16 < 32)
    (300, 8, 40, 24, 0, 64, False, 1),      # H = 8 < 16, 24 input channels (one partial chunk), W tail
    (5, 120, 130, 128, 0, 192, True, 9),    # ... with statistics: never staged (3 does not divide 32), per-tap kernels
    (1, 64, 64, 128, 0, 128, True, 9),      # few tiles: per-tap kernels
]


@pytest.mark.parametrize("case", STAGE_CASES)
def test_conv3x3_staged_kernel(case):
    """Launches with many tiles (the upper levels of the production step) run on conv3x3_stage_kernel: one barrier per
    32-channel chunk, operands by buffer loads with zero fill, deferred / carried epilogues.  Against torch on bf16-rounded
    operands, same tolerances as the other bf16 launches."""
    from stain2stain_amd import ops
    B, H, W, c0, c1, cout, stats, level = case
    mode = int(os.environ.get("S2S_CONV_STAGE", "1"))
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(21)
    x = torch.rand(B, c0 + c1, H, W, generator=g) * 2 - 1
    w = (torch.rand(cout, c0 + c1, 3, 3, generator=g) * 2 - 1) * 0.1
    ref = F.conv2d(rnd(x, dtype), rnd(w, dtype), None, padding=1)
    xs = nhwc(x, dtype)
    x0, x1 = xs[..., :c0], (xs[..., c0:] if c1 else None)
    wf, _ = ops.pack_conv3x3(w.to(DEV), dtype)
    path = ops._L().s2s_conv3x3_staged(0, B, H, W, cout, c0, c1, xs.shape[3], xs.shape[3], cout, int(stats))
    assert path == ((2 if c0 + c1 <= 64 else 1) if mode >= level else 0)
    if stats and path:                                 # staged: one row per (workgroup, wave row), 256 x 8
        assert ops._L().s2s_conv3x3_stat_rows(0, B, H, W, cout, c0, c1, xs.shape[3], xs.shape[3], 0) == 256 * 8
    y, stat = ops.conv3x3(x0, x1, wf, None, cout, want_stats=stats)
    torch.cuda.synchronize()
    assert relerr(nchw(y), ref) < tol_act(dtype)
    if stats:
        yb = nchw(y)                                   # statistics are taken of the values as stored
        s = stat.sum(-1).cpu()
        assert relerr(s[0], yb.sum((0, 2, 3))) < 2e-4
        assert relerr(s[1], (yb * yb).sum((0, 2, 3))) < 2e-4


@pytest.mark.parametrize("case", [(6, 250, 270, 64, 0, 64), (3, 250, 270, 64, 0, 136), (16, 100, 130, 64, 128, 64),
                                  (6, 250, 270, 8, 0, 64)])
def test_conv3x3_staged_eval_affine(case):
    """The sampler's launches (eval mode: conv bias + folded BatchNorm + ReLU in the epilogue) on the staged kernel's
    affine form: resident with one / three channel tiles, streaming from two sources, the 8-channel stem image."""
    from stain2stain_amd import ops
    B, H, W, c0, c1, cout = case
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(23)
    x = torch.rand(B, c0 + c1, H, W, generator=g) * 2 - 1
    w = (torch.rand(cout, c0 + c1, 3, 3, generator=g) * 2 - 1) * 0.1
    b = torch.rand(cout, generator=g) - 0.5
    sc, sh = torch.rand(cout, generator=g) + 0.5, torch.rand(cout, generator=g) - 0.5
    ref = ((F.conv2d(rnd(x, dtype), rnd(w, dtype), b, padding=1)) * sc[None, :, None, None] + sh[None, :, None, None]).clamp_min(0)
    xs = nhwc(x, dtype)
    x0, x1 = xs[..., :c0], (xs[..., c0:] if c1 else None)
    wf, _ = ops.pack_conv3x3(w.to(DEV), dtype)
    if int(os.environ.get("S2S_CONV_STAGE", "1")) >= 1:
        assert ops._L().s2s_conv3x3_staged(0, B, H, W, cout, c0, c1, xs.shape[3], xs.shape[3], cout, 0) > 0
    y, _ = ops.conv3x3(x0, x1, wf, b.to(DEV), cout, scale=sc.to(DEV), shift=sh.to(DEV), relu=True)
    assert relerr(nchw(y), ref) < tol_act(dtype)
    y, _ = ops.conv3x3(x0, x1, wf, None, cout, scale=sc.to(DEV), shift=sh.to(DEV), relu=False)      # no bias, no ReLU
    ref2 = F.conv2d(rnd(x, dtype), rnd(w, dtype), None, padding=1) * sc[None, :, None, None] + sh[None, :, None, None]
    assert relerr(nchw(y), ref2) < tol_act(dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3x3_dgrad(case, dtype):
    """Data gradient = the same kernel on the flipped/transposed packing."""
    from stain2stain_amd import ops
    B, H, W, c0, c1, cout = case
    cin = c0 + c1
    g = torch.Generator().manual_seed(9)
    w = (torch.rand(cout, cin, 3, 3, generator=g) * 2 - 1) * 0.1
    dy = torch.rand(B, cout, H, W, generator=g) * 2 - 1
    x = torch.zeros(B, cin, H, W, requires_grad=True)
    (F.conv2d(x, rnd(w, dtype), None, padding=1) * rnd(dy, dtype)).sum().backward()
    _, wd = ops.pack_conv3x3(w.to(DEV), dtype)
    dx, _ = ops.conv3x3(nhwc(dy, dtype), None, wd, None, cin)
    assert relerr(nchw(dx), x.grad) < tol_act(dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3x3_wgrad(case, dtype):
    from stain2stain_amd import ops
    B, H, W, c0, c1, cout = case
    cin = c0 + c1
    g = torch.Generator().manual_seed(10)
    x = torch.rand(B, cin, H, W, generator=g) * 2 - 1
    dy = torch.rand(B, cout, H, W, generator=g) * 2 - 1
    w = torch.zeros(cout, cin, 3, 3, requires_grad=True)
    (F.conv2d(rnd(x, dtype), w, None, padding=1) * rnd(dy, dtype)).sum().backward()
    xs = nhwc(x, dtype)
    grad = torch.full((cout, cin, 3, 3), 7.0, device=DEV)
    ops.conv3x3_wgrad(nhwc(dy, dtype), xs[..., :c0], xs[..., c0:] if c1 else None, grad, accumulate=False)
    assert relerr(grad.cpu(), w.grad) < 1e-4
    ops.conv3x3_wgrad(nhwc(dy, dtype), xs[..., :c0], xs[..., c0:] if c1 else None, grad, accumulate=True)
    assert relerr(grad.cpu(), 2 * w.grad) < 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cin,hc", [(3, 3), (1, 1), (4, 5), (6, 8)])
def test_stem_and_head(dtype, cin, hc):
    """cin 4 = RGB + mask channel (conditional_flow_matching_conditional_mask.py:62-64); hc = head channels
    (3 RGB velocity, 1 mask logit, num_classes logits)."""
    from stain2stain_amd import ops
    g = torch.Generator().manual_seed(11)
    x = torch.rand(2, cin, 21, 30, generator=g) * 2 - 1
    w = (torch.rand(16, cin, 3, 3, generator=g) * 2 - 1) * 0.3
    b = torch.rand(16, generator=g) - 0.5
    ref = F.conv2d(rnd(x, dtype), rnd(w, dtype), b, padding=1)   # bf16 mode stages image patch and weights in bf16
    y, stat = ops.stem_fwd(x.to(DEV), w.to(DEV), b.to(DEV), dtype)
    assert relerr(nchw(y), ref) < tol_act(dtype)
    s = stat.sum(-1).cpu()                 # [2][C][producer workgroups]
    assert relerr(s[0], ref.sum((0, 2, 3))) < 1e-4
    assert relerr(s[1], (ref * ref).sum((0, 2, 3))) < 1e-4
    # stem weight/bias gradient
    dy = torch.rand(2, 16, 21, 30, generator=g) * 2 - 1
    wv = w.clone().requires_grad_(True)
    bv = b.clone().requires_grad_(True)
    # the MFMA stem weight-gradient stages the image patch in the compute dtype (bf16 mode rounds it)
    (F.conv2d(rnd(x, dtype), wv, bv, padding=1) * rnd(dy, dtype)).sum().backward()
    dw = torch.empty(16, cin, 3, 3, device=DEV)
    db = torch.empty(16, device=DEV)
    ops.stem_wgrad(nhwc(dy, dtype), x.to(DEV), dw, db)
    assert relerr(dw.cpu(), wv.grad) < 1e-4
    assert relerr(db.cpu(), bv.grad) < 1e-4
    # head 1x1
    a = torch.rand(2, 16, 21, 30, generator=g) * 2 - 1
    hw = ((torch.rand(hc, 16, 1, 1, generator=g) * 2 - 1) * 0.3).requires_grad_(True)
    hb = (torch.rand(hc, generator=g) - 0.5).requires_grad_(True)
    av = rnd(a, dtype).requires_grad_(True)
    out = F.conv2d(av, hw, hb)
    v = ops.head_fwd(nhwc(a, dtype), hw.detach().to(DEV), hb.detach().to(DEV))
    assert relerr(v.cpu(), out) < 1e-5
    dv = torch.rand(out.shape, generator=g) - 0.5
    (out * dv).sum().backward()
    dw = torch.empty(hc, 16, 1, 1, device=DEV)
    db = torch.empty(hc, device=DEV)
    dx = ops.head_bwd(dv.to(DEV), nhwc(a, dtype), hw.detach().to(DEV), dw, db)
    assert relerr(nchw(dx), av.grad) < tol_act(dtype)
    assert relerr(dw.cpu(), hw.grad) < 1e-4
    assert relerr(db.cpu(), hb.grad) < 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("hw", [(12, 10), (13, 9)])
def test_bn_relu_pool_forward_backward(dtype, hw):
    """conv-output -> BN(train) -> ReLU -> {skip gradient, MaxPool2d(2)} against autograd on the CPU."""
    from oracle import unet_oracle as O
    from stain2stain_amd import ops
    H, W = hw
    B, C = 2, 16
    g = torch.Generator().manual_seed(12)
    z = rnd(torch.rand(B, C, H, W, generator=g) * 4 - 2, dtype).requires_grad_(True)
    gamma = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    beta = (torch.rand(C, generator=g) - 0.5).requires_grad_(True)
    y, mean, var = O.batchnorm_train(z, gamma, beta)
    y = y.clamp_min(0)
    if dtype == torch.bfloat16:   # the kernel stores y in bf16 and pools the stored value
        y = y + (y.detach().to(dtype).float() - y.detach())
    p = O.maxpool2(y)
    g1 = rnd(torch.rand(y.shape, generator=g) - 0.5, dtype)
    gp = rnd(torch.rand(p.shape, generator=g) - 0.5, dtype)
    ((y * g1).sum() + (p * gp).sum()).backward()

    zs = nhwc(z.detach(), dtype)
    n = B * H * W
    stat = torch.stack([zs.float().sum((0, 1, 2)), (zs.float() ** 2).sum((0, 1, 2))])[..., None].contiguous()
    rm = torch.zeros(C, device=DEV); rv = torch.ones(C, device=DEV); nb = torch.zeros((), dtype=torch.int64, device=DEV)
    st = ops.bn_finalize(stat, n, gamma.detach().to(DEV), beta.detach().to(DEV), rm, rv, nb)
    assert relerr(st[0].cpu(), mean) < 1e-5
    assert relerr(st[1].cpu(), torch.rsqrt(var + 1e-5)) < 1e-5
    assert relerr(rm.cpu(), 0.1 * mean) < 1e-5
    assert relerr(rv.cpu(), 0.9 + 0.1 * var * n / (n - 1)) < 1e-5
    assert int(nb) == 1
    ya, pool = ops.bn_relu_apply(zs, st[2], st[3], want_pool=True)
    assert relerr(nchw(ya), y) < tol_act(dtype)
    assert relerr(nchw(pool), p) < tol_act(dtype)
    dgam = torch.empty(C, device=DEV); dbet = torch.empty(C, device=DEV); dbias = torch.empty(C, device=DEV)
    dz = ops.bn_relu_bwd(nhwc(g1, dtype), nhwc(gp, dtype), zs, st, gamma.detach().to(DEV), dgam, dbet, dbias)
    assert relerr(nchw(dz), z.grad) < tol_act(dtype) * 2
    assert relerr(dgam.cpu(), gamma.grad) < 5e-3 if dtype == torch.bfloat16 else relerr(dgam.cpu(), gamma.grad) < 1e-4
    assert relerr(dbet.cpu(), beta.grad) < 5e-3 if dtype == torch.bfloat16 else relerr(dbet.cpu(), beta.grad) < 1e-4
    assert float(dbias.abs().max()) < 1e-2 * float(z.grad.abs().sum((0, 2, 3)).max())  # analytically zero


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_upsample_fwd_bwd_with_pad_and_bias(dtype):
    from oracle import unet_oracle as O
    from stain2stain_amd import ops
    g = torch.Generator().manual_seed(13)
    B, C, h, w = 2, 16, 5, 6
    Ho, Wo = 11, 13
    x = rnd(torch.rand(B, C, h, w, generator=g) * 2 - 1, dtype).requires_grad_(True)
    bias = (torch.rand(B, C, generator=g) - 0.5).requires_grad_(True)
    up = O.upsample2x_bilinear_ac(x + bias[:, :, None, None])
    up = F.pad(up, [(Wo - 2 * w) // 2, Wo - 2 * w - (Wo - 2 * w) // 2, (Ho - 2 * h) // 2, Ho - 2 * h - (Ho - 2 * h) // 2])
    gy = rnd(torch.rand(up.shape, generator=g) - 0.5, dtype)
    (up * gy).sum().backward()
    buf = torch.zeros(B, Ho, Wo, C + 8, device=DEV, dtype=dtype)
    ops.upsample2x_fwd(nhwc(x.detach(), dtype), buf[..., 8:], bias.detach().to(DEV))
    assert relerr(nchw(buf[..., 8:]), up) < tol_act(dtype)
    assert float(buf[..., :8].abs().max()) == 0.0
    gbuf = torch.zeros(B, Ho, Wo, C + 8, device=DEV, dtype=dtype)
    gbuf[..., 8:] = nhwc(gy, dtype)
    dx = ops.upsample2x_bwd(gbuf[..., 8:], h, w)
    assert relerr(nchw(dx), x.grad) < tol_act(dtype)
    assert relerr(ops.pixel_sum(dx).cpu(), bias.grad) < tol_act(dtype)


def test_time_path_and_loss():
    from oracle import unet_oracle as O
    from stain2stain_amd import ops
    g = torch.Generator().manual_seed(14)
    t = torch.rand(5, generator=g)
    assert relerr(ops.time_embedding(t.to(DEV), 32).cpu(), O.time_embedding(t, 32)) < 1e-6
    assert relerr(ops.time_embedding(t.to(DEV), 256).cpu(), O.time_embedding(t, 256)) < 1e-6
    x = torch.rand(5, 32, generator=g, requires_grad=True)
    w = (torch.rand(48, 32, generator=g) - 0.5).requires_grad_(True)
    b = (torch.rand(48, generator=g) - 0.5).requires_grad_(True)
    h = x @ w.t() + b
    a = h * torch.sigmoid(h)
    da = torch.rand(a.shape, generator=g) - 0.5
    (a * da).sum().backward()
    hg = ops.linear_fwd(x.detach().to(DEV), w.detach().to(DEV), b.detach().to(DEV))
    assert relerr(hg.cpu(), h) < 1e-6
    ag = ops.silu_fwd(hg)
    assert relerr(ag.cpu(), a) < 1e-6
    dh = ops.silu_bwd(hg, da.to(DEV))
    dw = torch.empty(48, 32, device=DEV); db = torch.empty(48, device=DEV)
    dx = ops.linear_bwd(dh, x.detach().to(DEV), w.detach().to(DEV), dw, db)
    assert relerr(dx.cpu(), x.grad) < 1e-5
    assert relerr(dw.cpu(), w.grad) < 1e-5
    assert relerr(db.cpu(), b.grad) < 1e-5
    # probability path + loss
    x0 = torch.rand(3, 3, 8, 12, generator=g) * 2 - 1
    x1 = torch.rand(3, 3, 8, 12, generator=g) * 2 - 1
    tt = torch.rand(3, generator=g)
    xt_ref, ut_ref = O.cfm_sample(x0, x1, tt)
    xt, ut = ops.cfm_sample(x0.to(DEV), x1.to(DEV), tt.to(DEV))
    assert relerr(xt.cpu(), xt_ref) < 1e-6 and relerr(ut.cpu(), ut_ref) < 1e-6
    v = torch.rand(x0.shape, generator=g, requires_grad=True)
    loss_ref = O.cfm_loss(v, ut_ref)
    loss_ref.backward()
    loss, dv = ops.mse_loss(v.detach().to(DEV), ut)
    assert relerr(loss.cpu(), loss_ref) < 1e-6
    assert relerr(dv.cpu(), v.grad) < 1e-6


def test_adam_matches_torch_optim():
    from stain2stain_amd import ops
    g = torch.Generator().manual_seed(15)
    p = torch.nn.Parameter(torch.rand(1000, generator=g) - 0.5)
    opt = torch.optim.Adam([p], lr=1e-3, weight_decay=1e-2)
    pg = p.detach().clone().to(DEV)
    m = torch.zeros_like(pg); v = torch.zeros_like(pg)
    for step in range(1, 4):
        gr = torch.rand(1000, generator=g) - 0.5
        p.grad = gr.clone()
        opt.step()
        ops.adam_step_(pg, gr.to(DEV), m, v, step, 1e-3, weight_decay=1e-2)
        assert relerr(pg.cpu(), p.detach()) < 1e-6


def test_layout_roundtrip_and_error_status():
    from stain2stain_amd import ops
    x = torch.rand(2, 16, 5, 7)
    for dt in (torch.float32, torch.bfloat16):
        y = ops.nchw_to_nhwc(x.to(DEV), dt)
        assert relerr(nchw(y), rnd(x, dt)) == 0.0
        assert relerr(ops.nhwc_to_nchw(y).cpu(), rnd(x, dt)) == 0.0
    with pytest.raises(RuntimeError):   # channel count not a multiple of 8 -> negative status -> exception
        ops.nchw_to_nhwc(torch.rand(1, 12, 4, 4, device=DEV), torch.float32)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C", [8, 16, 64, 24])
def test_head_loss_fused_matches_separate_ops(dtype, C):
    """Fused head conv + MSE + backward (training step) against autograd on the CPU.  C = 8, 16, 64 take the
    lane-per-piece kernel (C/8 a power of two), C = 24 the one-thread-per-pixel form."""
    from stain2stain_amd import ops
    g = torch.Generator().manual_seed(21)
    B, H, W = 2, 19, 23
    a = rnd(torch.rand(B, C, H, W, generator=g) * 2 - 1, dtype).requires_grad_(True)
    w = ((torch.rand(3, C, 1, 1, generator=g) * 2 - 1) * 0.3).requires_grad_(True)
    b = (torch.rand(3, generator=g) - 0.5).requires_grad_(True)
    u = torch.rand(B, 3, H, W, generator=g) * 2 - 1
    v_ref = F.conv2d(a, w, b)
    loss_ref = ((v_ref - u) ** 2).mean()
    loss_ref.backward()
    dw = torch.empty(3, C, 1, 1, device=DEV); db = torch.empty(3, device=DEV)
    loss, dx, v = ops.head_loss_fused(nhwc(a.detach(), dtype), w.detach().to(DEV), b.detach().to(DEV), u.to(DEV), dw, db,
                                      want_v=True)
    assert relerr(v.cpu(), v_ref) < 1e-5
    assert relerr(loss.cpu(), loss_ref) < 1e-5
    assert relerr(nchw(dx), a.grad) < tol_act(dtype)
    assert relerr(dw.cpu(), w.grad) < (5e-3 if C == 24 else 1e-4)   # the per-pixel form stages a bf16 copy for dW
    assert relerr(db.cpu(), b.grad) < 1e-4


def test_timing_only_events_bracket_a_kernel():
    """s2s_event_*: the timestamp-only events of bench.py's per-kernel brackets (no system-scope fence) measure the same
    stream interval as torch's events around the same launches, to well within a launch time."""
    from stain2stain_amd import ops
    x = torch.zeros(1 << 24, device="cuda")
    y = torch.ones(1 << 24, device="cuda")
    for _ in range(3):
        ops.axpy_(x, y, 1.0)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0, e1 = ops.TimingEvent(), ops.TimingEvent()
    t0.record(); e0.record()
    for _ in range(20):
        ops.axpy_(x, y, 1.0)
    e1.record(); t1.record()
    torch.cuda.synchronize()
    ms, ms_torch = e0.elapsed_time(e1), t0.elapsed_time(t1)
    assert 0.0 < ms <= ms_torch * 1.05 + 0.02 and ms > 0.5 * ms_torch, (ms, ms_torch)
    assert float(x[0]) == 23.0


def test_cu_masked_stream_runs_kernels():
    """s2s_stream_create_cu_mask: a stream confined to three quarters of the CUs runs a conv launch to the same bits as
    the default stream (engine.run_on_side's stream when S2S_WGRAD_CUS is set); bad specs are refused.  In process: the
    masked hipStream_t is cached for the life of the process and never destroyed (ops.cu_masked_stream), so torch's
    caching allocator -- which keys the blocks allocated under the stream by its raw handle -- never sees a dangling one
    (round 2 destroyed it when the wrapper was collected and the suite took a segmentation fault at interpreter exit)."""
    import gc
    from stain2stain_amd import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    x = (torch.rand(2, 32, 32, 64, device="cuda", generator=g) - 0.5).to(torch.bfloat16)
    w = (torch.rand(64, 64, 3, 3, device="cuda", generator=g) - 0.5) * 0.1
    wf, _ = ops.pack_conv3x3(w, torch.bfloat16)
    ref, _ = ops.conv3x3(x, None, wf, None, 64)
    side = ops.cu_masked_stream("cuda:0", "3:4")
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        got, _ = ops.conv3x3(x, None, wf, None, 64)
    side.synchronize()
    assert torch.equal(got, ref)
    assert ops.cu_masked_stream("cuda:0", "3:4") is side          # one queue per (device, K, M)
    handle = side.cuda_stream
    del got, side
    gc.collect()
    torch.cuda.empty_cache()                                       # the allocator lets go of the blocks of that stream
    again = ops.cu_masked_stream("cuda:0", "3:4")
    assert again.cuda_stream == handle
    with torch.cuda.stream(again):
        got, _ = ops.conv3x3(x, None, wf, None, 64)
    again.synchronize()
    assert torch.equal(got, ref)
    for bad in ("0:4", "5:4"):
        with pytest.raises(ValueError):
            ops.cu_masked_stream("cuda:0", bad)
