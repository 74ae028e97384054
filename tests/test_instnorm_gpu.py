"""Fused InstanceNorm2d + LeakyReLU on the GPU against torch's own operator pair in fp32
(what pins this row: the reference has no pix2pix code).  Tolerances as in test_ops_gpu.py: fp32 mode 1e-4 (north_star
asks 1e-3), bf16 storage 4e-3 on element-wise outputs (one bf16 rounding is 2^-9), 1e-3 on the fp32 affine gradients."""
import pytest
import torch
import torch.nn.functional as F

from conftest import relerr

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _case(B, C, H, W, affine, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, H, W, generator=g) * 1.7 + 0.4
    gamma = torch.rand(C, generator=g) + 0.5 if affine else None
    beta = torch.rand(C, generator=g) - 0.5 if affine else None
    cot = torch.rand(B, C, H, W, generator=g) - 0.5
    return x, gamma, beta, cot


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("affine", [False, True])
@pytest.mark.parametrize("slope", [0.2, 0.0])
@pytest.mark.parametrize("shape", [(2, 16, 9, 11), (3, 72, 4, 4), (1, 264, 13, 2), (2, 8, 1, 2)])
def test_ops_against_torch(shape, slope, affine, dtype):
    from stain2stain_amd import ops
    B, C, H, W = shape
    x, gamma, beta, cot = _case(B, C, H, W, affine, 7 + C)
    rb = (lambda a: a.to(dtype).float())
    xr = rb(x).requires_grad_(True)
    gr = None if gamma is None else gamma.clone().requires_grad_(True)
    br = None if beta is None else beta.clone().requires_grad_(True)
    ref = F.leaky_relu(F.instance_norm(xr, weight=gr, bias=br, eps=1e-5), slope)
    (ref * rb(cot)).sum().backward()
    # the operands as channel slices of wider NHWC buffers (exercises the pixel stride)
    xb = torch.zeros(B, H, W, C + 8, device=DEV, dtype=dtype); xb[..., 8:] = x.permute(0, 2, 3, 1).to(DEV, dtype)
    gb = torch.zeros(B, H, W, C + 16, device=DEV, dtype=dtype); gb[..., :C] = cot.permute(0, 2, 3, 1).to(DEV, dtype)
    y, stats = ops.instnorm_lrelu_fwd(xb[..., 8:], None if gamma is None else gamma.to(DEV),
                                      None if beta is None else beta.to(DEV), 1e-5, slope)
    tol = 4e-3 if dtype == torch.bfloat16 else 1e-4
    assert relerr(y.float().cpu().permute(0, 3, 1, 2), ref) < tol
    assert relerr(stats[0].cpu(), xr.detach().mean((2, 3))) < 1e-5 * max(1.0, float(xr.detach().abs().max()))
    dg = torch.full((C,), 3.0, device=DEV) if affine else None
    db = torch.full((C,), 3.0, device=DEV) if affine else None
    dx = ops.instnorm_lrelu_bwd(gb[..., :C], xb[..., 8:], stats, slope, dg, db)
    assert relerr(dx.float().cpu().permute(0, 3, 1, 2), xr.grad) < 2 * tol
    if affine:
        assert relerr(dg.cpu(), gr.grad) < 1e-3 and relerr(db.cpu(), br.grad) < 1e-3
        ops.instnorm_lrelu_bwd(gb[..., :C], xb[..., 8:], stats, slope, dg, db, accumulate=True)
        assert relerr(dg.cpu(), 2 * gr.grad) < 1e-3 and relerr(db.cpu(), 2 * br.grad) < 1e-3


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_module_matches_torch_pair_through_autograd(precision):
    from stain2stain_amd import InstanceNormLeakyReLU
    x, gamma, beta, cot = _case(2, 32, 12, 10, True, 21)
    m = InstanceNormLeakyReLU(32, affine=True, negative_slope=0.2, precision=precision).to(DEV)
    ref = torch.nn.Sequential(torch.nn.InstanceNorm2d(32, affine=True), torch.nn.LeakyReLU(0.2))
    sd = {"weight": gamma, "bias": beta}
    m.load_state_dict(sd); ref[0].load_state_dict(sd)
    xr = (x.to(torch.bfloat16).float() if precision == "bf16" else x.clone()).requires_grad_(True)
    (ref(xr) * cot).sum().backward()
    xg = x.to(DEV).requires_grad_(True)                                   # plain NCHW fp32 input: converted on the way in
    out = m(xg)
    (out.float() * cot.to(DEV)).sum().backward()
    tol = 8e-3 if precision == "bf16" else 1e-4
    assert relerr(out.float().cpu(), ref(xr).detach()) < tol
    assert relerr(xg.grad.cpu(), xr.grad) < 2 * tol
    assert relerr(m.weight.grad.cpu(), ref[0].weight.grad) < 5e-3 and relerr(m.bias.grad.cpu(), ref[0].bias.grad) < 5e-3
    with pytest.raises(ValueError):
        m(torch.rand(2, 32, 1, 1, device=DEV))


def test_pix2pix_level_size_bf16():
    """A generator-level tensor at BASELINE.json's headline configuration (batch 16, 128 channels at 128x128, bf16)
    against torch's operator pair in fp32 on the same device; input gradient in L2 (a LeakyReLU decision within
    rounding of zero may differ for a handful of the 33 M elements)."""
    from stain2stain_amd import ops
    g = torch.Generator(device=DEV).manual_seed(1984)
    xs = (torch.randn(16, 128, 128, 128, device=DEV, generator=g) * 1.5 + 0.3).to(torch.bfloat16)
    cot = (torch.rand(16, 128, 128, 128, device=DEV, generator=g) - 0.5).to(torch.bfloat16)
    xr = xs.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    ref = F.leaky_relu(F.instance_norm(xr, eps=1e-5), 0.2)
    (ref * cot.float().permute(0, 3, 1, 2)).sum().backward()
    y, stats = ops.instnorm_lrelu_fwd(xs, None, None, 1e-5, 0.2)
    assert relerr(y.float().permute(0, 3, 1, 2), ref) < 4e-3
    dx = ops.instnorm_lrelu_bwd(cot, xs, stats, 0.2)
    d, r = dx.float().permute(0, 3, 1, 2).double(), xr.grad.double()
    assert float((d - r).norm() / r.norm()) < 3e-3
