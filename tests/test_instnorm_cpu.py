"""Row a13 building block on the CPU: the oracle's InstanceNorm + LeakyReLU formulas against torch's own operators
(the only pin this row has: the reference repository holds no pix2pix code), and the module's host-side contract."""
import pytest
import torch
import torch.nn.functional as F

from conftest import relerr


@pytest.mark.parametrize("affine", [False, True])
@pytest.mark.parametrize("slope", [0.2, 0.0])
def test_oracle_matches_torch_instance_norm_leaky_relu(affine, slope):
    from oracle import pix2pix_oracle as P
    g = torch.Generator().manual_seed(1984)
    x = (torch.randn(3, 16, 9, 11, generator=g) * 2 + 0.5).double().requires_grad_(True)
    gamma = (torch.rand(16, generator=g) + 0.5).double().requires_grad_(True) if affine else None
    beta = (torch.rand(16, generator=g) - 0.5).double().requires_grad_(True) if affine else None
    cot = torch.randn(3, 16, 9, 11, generator=g).double()
    ref = F.leaky_relu(F.instance_norm(x, weight=gamma, bias=beta, eps=1e-5), slope)
    got = P.instance_norm_lrelu(x.detach(), None if gamma is None else gamma.detach(),
                                None if beta is None else beta.detach(), 1e-5, slope)
    assert relerr(got, ref) < 1e-12
    (ref * cot).sum().backward()
    dx, dg, db = P.instance_norm_lrelu_grads(x.detach(), None if gamma is None else gamma.detach(),
                                             None if beta is None else beta.detach(), cot, 1e-5, slope)
    assert relerr(dx, x.grad) < 1e-10
    if affine:
        assert relerr(dg, gamma.grad) < 1e-10 and relerr(db, beta.grad) < 1e-10


def test_module_contract_without_a_gpu():
    from stain2stain_amd import InstanceNormLeakyReLU
    m = InstanceNormLeakyReLU(64, affine=True)
    ref = torch.nn.InstanceNorm2d(64, affine=True)
    assert set(m.state_dict()) == set(ref.state_dict())                 # weight, bias; no running statistics
    assert set(InstanceNormLeakyReLU(64).state_dict()) == set(torch.nn.InstanceNorm2d(64).state_dict()) == set()
    with pytest.raises(RuntimeError):                                   # HIP-only: no silent CPU fallback
        m(torch.rand(2, 64, 8, 8))
    with pytest.raises(ValueError):
        InstanceNormLeakyReLU(12)
