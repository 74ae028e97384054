"""pix2pix generator + PatchGAN discriminator assembled from the HIP layers, against the
same networks built from torch's own layers in fp32 on the CPU (builder-authored oracle: the reference has no such
model, so this row is parity-UNPINNED with respect to it).  The HIP path stores activations in bf16 (there is no fp32
mode for these kernels yet) and InstanceNorm over the 2x2 / 4x4 maps of the inner levels amplifies that rounding, so the
bounds are the loose ones of a bf16 pipeline: outputs in L2, losses to a few per cent, gradients by direction."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _l2(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).norm() / b.norm())


def _pair(seed=1984):
    from oracle import pix2pix_oracle as O
    from stain2stain_amd.pix2pix import PatchGANDiscriminator, Pix2PixGenerator
    torch.manual_seed(seed)
    G, D = Pix2PixGenerator(ngf=16, num_downs=6), PatchGANDiscriminator(ndf=16)
    Go, Do = O.OracleGenerator(ngf=16, num_downs=6), O.OracleDiscriminator(ndf=16)
    rb = lambda t: t.to(torch.bfloat16).float()
    sd_g = {k: rb(v) for k, v in G.state_dict().items()}
    sd_d = {k: rb(v) for k, v in D.state_dict().items()}
    G.load_state_dict(sd_g); D.load_state_dict(sd_d)
    assert set(Go.state_dict()) == set(sd_g) and set(Do.state_dict()) == set(sd_d)       # same parameter names
    Go.load_state_dict(sd_g); Do.load_state_dict(sd_d)
    return G.to(DEV), D.to(DEV), Go, Do


def test_networks_match_the_torch_layer_oracle():
    from oracle.pix2pix_oracle import pix2pix_losses
    G, D, Go, Do = _pair()
    g = torch.Generator().manual_seed(7)
    src = (torch.rand(4, 3, 64, 64, generator=g) * 2 - 1).to(torch.bfloat16).float()
    tgt = (torch.rand(4, 3, 64, 64, generator=g) * 2 - 1).to(torch.bfloat16).float()
    fake, ld, lg = pix2pix_losses(G, D, src.to(DEV), tgt.to(DEV))
    fake_o, ld_o, lg_o = pix2pix_losses(Go, Do, src, tgt)
    assert fake.shape == (4, 3, 64, 64) and D(src.to(DEV), tgt.to(DEV)).shape == (4, 1, 6, 6)
    print(f"G output rel-L2 {_l2(fake.cpu(), fake_o):.3e}; loss_D {float(ld.detach()):.5f} vs {float(ld_o.detach()):.5f}; "
          f"loss_G {float(lg.detach()):.4f} vs {float(lg_o.detach()):.4f}")
    assert _l2(fake.cpu(), fake_o) < 3e-2                       # measured 6.5e-3
    ldf, lgf, ldo, lgo = (float(v.detach()) for v in (ld, lg, ld_o, lg_o))
    assert abs(ldf - ldo) < 2e-3 * abs(ldo) and abs(lgf - lgo) < 2e-3 * abs(lgo)
    (ld + lg).backward()
    (ld_o + lg_o).backward()
    worst = 1.0
    pairs = list(zip(list(G.named_parameters()) + list(D.named_parameters()),
                     list(Go.named_parameters()) + list(Do.named_parameters())))
    scale = max(float(q.grad.norm()) for _, (_, q) in pairs)
    for (k, p), (_, q) in pairs:
        a, b = p.grad.float().cpu(), q.grad
        if float(b.norm()) < 1e-4 * scale:
            continue                      # a conv bias ahead of InstanceNorm: its gradient is analytically zero (noise)
        cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
        worst = min(worst, cos)
        assert cos > 0.95, (k, cos)                             # measured >= 0.985
    print(f"smallest gradient cosine {worst:.4f}")


def test_g_plus_d_step_trains():
    """A few G + D steps on a fixed batch: finite losses, the L1-dominated generator loss goes down."""
    from oracle.pix2pix_oracle import pix2pix_step
    G, D, _, _ = _pair(3)
    og = torch.optim.Adam(G.parameters(), lr=2e-4, betas=(0.5, 0.999))
    od = torch.optim.Adam(D.parameters(), lr=2e-4, betas=(0.5, 0.999))
    g = torch.Generator().manual_seed(9)
    src = (torch.rand(4, 3, 64, 64, generator=g) * 2 - 1).to(DEV)
    tgt = (src * 0.5).clone()
    hist = [tuple(float(v) for v in pix2pix_step(G, D, og, od, src, tgt)) for _ in range(12)]
    assert all(l == l and abs(l) < 1e4 for pair in hist for l in pair)
    assert hist[-1][1] < hist[0][1]


def test_headline_networks_forward_against_the_oracle():
    """The bench configuration's networks (8-level generator with ngf = 64, PatchGAN with ndf = 64) on 256x256 tiles,
    batch 2, forward only, against the torch-layer oracle on the CPU: generator output and discriminator logits in L2."""
    import os
    from oracle import pix2pix_oracle as O
    from stain2stain_amd import PatchGANDiscriminator, Pix2PixGenerator
    torch.manual_seed(1984)
    G, D = Pix2PixGenerator(), PatchGANDiscriminator()
    rb = lambda t: t.to(torch.bfloat16).float()
    sd_g, sd_d = {k: rb(v) for k, v in G.state_dict().items()}, {k: rb(v) for k, v in D.state_dict().items()}
    G.load_state_dict(sd_g); D.load_state_dict(sd_d)
    Go, Do = O.OracleGenerator(), O.OracleDiscriminator()
    Go.load_state_dict(sd_g); Do.load_state_dict(sd_d)
    g = torch.Generator().manual_seed(11)
    src = rb(torch.rand(2, 3, 256, 256, generator=g) * 2 - 1)
    tgt = rb(torch.rand(2, 3, 256, 256, generator=g) * 2 - 1)
    G, D = G.to(DEV), D.to(DEV)
    with torch.no_grad():
        fake = G(src.to(DEV)).cpu()
        logits = D(src.to(DEV), tgt.to(DEV)).cpu()
        threads = torch.get_num_threads()
        torch.set_num_threads(min(16, os.cpu_count() or 8))
        try:
            fake_o, logits_o = Go(src), Do(src, tgt)
        finally:
            torch.set_num_threads(threads)
    assert fake.shape == (2, 3, 256, 256) and logits.shape == (2, 1, 30, 30)
    print(f"headline G output rel-L2 {_l2(fake, fake_o):.3e}, D logits rel-L2 {_l2(logits, logits_o):.3e}")
    assert _l2(fake, fake_o) < 3e-2 and _l2(logits, logits_o) < 3e-2      # measured 7.0e-3 / 6.0e-3


def test_module_face_fp32_mode_matches_the_oracle_and_runs_on_the_engine_passes():
    """``G(x)`` / ``D(a, b)`` under autograd are one node each over the fused engine's passes (pix2pix_engine.NetRunner):
    in the fp32 parity mode the outputs, both losses and every gradient follow the torch-layer oracle at 1e-3 (gradients
    relative to the largest gradient norm of the network); an optimiser step re-packs the MFMA operands."""
    from oracle import pix2pix_oracle as O
    from stain2stain_amd.pix2pix import PatchGANDiscriminator, Pix2PixGenerator, _DiscriminatorFn, _GeneratorFn
    torch.manual_seed(5)
    G, D = Pix2PixGenerator(ngf=16, num_downs=5, precision="fp32"), PatchGANDiscriminator(ndf=16, precision="fp32")
    Go, Do = O.OracleGenerator(ngf=16, num_downs=5), O.OracleDiscriminator(ndf=16)
    Go.load_state_dict(G.state_dict()); Do.load_state_dict(D.state_dict())
    G, D = G.to(DEV), D.to(DEV)
    g = torch.Generator().manual_seed(17)
    src, tgt = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1, torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
    fake, ld, lg = O.pix2pix_losses(G, D, src.to(DEV), tgt.to(DEV))
    fake_o, ld_o, lg_o = O.pix2pix_losses(Go, Do, src, tgt)
    assert type(fake.grad_fn).__name__ == _GeneratorFn.__name__ + "Backward"
    assert type(D(src.to(DEV), tgt.to(DEV)).grad_fn).__name__ == _DiscriminatorFn.__name__ + "Backward"
    assert float((fake.detach().cpu() - fake_o.detach()).abs().max()) < 1e-3 * float(fake_o.detach().abs().max())
    ldv, lgv, ldo, lgo = (float(v.detach()) for v in (ld, lg, ld_o, lg_o))
    assert abs(ldv - ldo) < 1e-3 * abs(ldo) and abs(lgv - lgo) < 1e-3 * abs(lgo)
    (ld + lg).backward()
    (ld_o + lg_o).backward()
    for mod, ref in ((G, Go), (D, Do)):
        scale = max(float(q.grad.norm()) for q in ref.parameters())
        for (k, p), (_, q) in zip(mod.named_parameters(), ref.named_parameters()):
            err = float((p.grad.cpu() - q.grad).norm())
            assert err < 3e-3 * scale, (k, err / scale)
    # a parameter update is seen by the next forward (packed operands are refreshed on a version change)
    before = G(src.to(DEV)).detach().clone()
    with torch.no_grad():
        for p in G.parameters():
            p.mul_(1.01)
    after = G(src.to(DEV)).detach()
    assert not torch.equal(before, after)
    with pytest.raises(RuntimeError, match="no gradient with respect to the input"):
        G(src.to(DEV).requires_grad_(True)).sum().backward()
