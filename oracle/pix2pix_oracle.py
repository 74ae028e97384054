"""CPU restatement of the pix2pix building blocks of SURVEY.md section 8, row a13.  TEST INFRASTRUCTURE ONLY: imported
by tests/ and __graft_entry__.smoke() (and nothing else); the product path never touches it.

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


# ---- the two networks, from torch's own layers (builder-authored: the reference has none) -------------
class OracleGenerator(torch.nn.Module):
    """Same architecture and parameter names as stain2stain_amd.pix2pix.Pix2PixGenerator, on nn.Conv2d /
    nn.ConvTranspose2d / nn.InstanceNorm2d in fp32."""

    def __init__(self, in_channels=3, out_channels=3, ngf=64, num_downs=8):
        super().__init__()
        nn = torch.nn
        ch = [ngf * min(2 ** i, 8) for i in range(num_downs)]
        self.out_channels = out_channels
        self.downs = nn.ModuleList([nn.Conv2d(8 if i == 0 else ch[i - 1], ch[i], 4, 2, 1) for i in range(num_downs)])
        ups = []
        for i in range(num_downs - 1, -1, -1):
            cin = ch[i] if i == num_downs - 1 else 2 * ch[i]
            ups.append(nn.ConvTranspose2d(cin, 8 if i == 0 else ch[i - 1], 4, 2, 1))
        self.ups = nn.ModuleList(ups)

    def forward(self, x):
        F = torch.nn.functional
        n = len(self.downs)
        x = F.pad(x, (0, 0, 0, 0, 0, 8 - x.shape[1]))
        acts = []
        h = F.leaky_relu(self.downs[0](x), 0.2)
        acts.append(h)
        for i in range(1, n - 1):
            h = F.leaky_relu(F.instance_norm(self.downs[i](h)), 0.2)
            acts.append(h)
        h = torch.relu(self.downs[n - 1](h))
        for j, up in enumerate(self.ups):
            h = up(h)
            if j < n - 1:
                h = torch.relu(F.instance_norm(h))
                h = torch.cat([torch.relu(acts[n - 2 - j]), h], 1)
        return torch.tanh(h[:, :self.out_channels])


class OracleDiscriminator(torch.nn.Module):
    """Same architecture and parameter names as stain2stain_amd.pix2pix.PatchGANDiscriminator (``n_layers = 3``: the
    70x70 PatchGAN c1 ... c5), on nn.Conv2d + F.instance_norm in fp32."""

    def __init__(self, in_channels=6, ndf=64, n_layers=3):
        super().__init__()
        nn = torch.nn
        self.n_layers = n_layers
        self.c1 = nn.Conv2d(8, ndf, 4, 2, 1)
        cin = ndf
        for n in range(1, n_layers + 1):
            cout = ndf * min(2 ** n, 8)
            setattr(self, f"c{n + 1}", nn.Conv2d(cin, cout, 4, 1 if n == n_layers else 2, 1))
            cin = cout
        setattr(self, f"c{n_layers + 2}", nn.Conv2d(cin, 8, 4, 1, 1))

    def forward(self, a, b):
        F = torch.nn.functional
        x = torch.cat([a, b], 1)
        x = F.pad(x, (0, 0, 0, 0, 0, 8 - x.shape[1]))
        h = F.leaky_relu(self.c1(x), 0.2)
        for k in range(2, self.n_layers + 2):
            h = F.leaky_relu(F.instance_norm(getattr(self, f"c{k}")(h)), 0.2)
        return getattr(self, f"c{self.n_layers + 2}")(h)[:, :1]


def pix2pix_losses(G, D, src, tgt, lambda_l1: float = 100.0):
    """One pix2pix evaluation on any pair of modules with the (x) -> fake / (a, b) -> logits signatures: (fake, loss_D,
    loss_G) with the vanilla GAN objective (BCE with logits) + lambda * L1 -- the definition the fused HIP step
    (stain2stain_amd.Pix2PixTrainer) is checked against."""
    bce = torch.nn.functional.binary_cross_entropy_with_logits
    fake = G(src)
    pr, pf = D(src, tgt), D(src, fake.detach())
    loss_d = 0.5 * (bce(pr, torch.ones_like(pr)) + bce(pf, torch.zeros_like(pf)))
    pg = D(src, fake)
    loss_g = bce(pg, torch.ones_like(pg)) + lambda_l1 * (fake - tgt).abs().mean()
    return fake, loss_d, loss_g


def pix2pix_step(G, D, opt_g, opt_d, src, tgt, lambda_l1: float = 100.0):
    """The G + D optimisation step of pix2pix with torch optimisers: discriminator update on (real, detached fake), then
    the generator update through the updated discriminator.  Returns (loss_D, loss_G) as tensors."""
    bce = torch.nn.functional.binary_cross_entropy_with_logits
    fake = G(src)
    opt_d.zero_grad(set_to_none=True)
    pr, pf = D(src, tgt), D(src, fake.detach())
    loss_d = 0.5 * (bce(pr, torch.ones_like(pr)) + bce(pf, torch.zeros_like(pf)))
    loss_d.backward()
    opt_d.step()
    opt_g.zero_grad(set_to_none=True)
    for p in D.parameters():
        p.requires_grad_(False)
    pg = D(src, fake)
    loss_g = bce(pg, torch.ones_like(pg)) + lambda_l1 * (fake - tgt).abs().mean()
    loss_g.backward()
    for p in D.parameters():
        p.requires_grad_(True)
    opt_g.step()
    return loss_d.detach(), loss_g.detach()
