"""Size-independent properties at the BASELINE.json production sizes (where the CPU oracle would take minutes).

* adjoint identity of the three convolution kernels on production layer shapes (batch 16):
      <conv(x; w), dy> = <x, dgrad(dy; w)> = <w, wgrad(x, dy)>
  Any indexing / tiling / masking error in one of the kernels breaks the equalities; the inner products are formed
  in fp64 by torch (test infrastructure).  bf16 mode: 5e-3 relative (each kernel rounds its bf16 output once,
  2^-9), fp32 mode: 2e-5.
* BatchNorm invariants of a full forward at 256x256: normalised activations have the per-channel moments the
  forward's own statistics imply.
* cfg5-sized tiles (512x512) run through the whole training step.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"

# (H, c0, c1, cout): the production U-Net's layer shapes at 256x256 input, batch 16
LAYERS = [(256, 64, 0, 64), (128, 64, 0, 128), (64, 256, 0, 256), (32, 512, 1024, 512), (16, 1024, 0, 1024),
          (256, 64, 128, 64)]


def _dot(a, b):
    return float((a.double() * b.double()).sum())


@pytest.mark.parametrize("layer", LAYERS)
def test_conv_adjoint_identity_bf16_production_shapes(layer):
    from stain2stain_amd import ops
    H, c0, c1, cout = layer
    B, cin, dt = 16, c0 + c1, torch.bfloat16
    g = torch.Generator(device=DEV).manual_seed(H * 7 + cout)
    x = (torch.rand(B, H, H, cin, device=DEV, generator=g) * 2 - 1).to(dt)
    dy = (torch.rand(B, H, H, cout, device=DEV, generator=g) * 2 - 1).to(dt)
    w = ((torch.rand(cout, cin, 3, 3, device=DEV, generator=g) * 2 - 1) * 0.05).to(dt).float()   # bf16-exact master
    wf, wd = ops.pack_conv3x3(w, dt)
    x0, x1 = x[..., :c0], (x[..., c0:] if c1 else None)
    y, _ = ops.conv3x3(x0, x1, wf, None, cout)
    dx, _ = ops.conv3x3(dy, None, wd, None, cin)
    dw = torch.empty(cout, cin, 3, 3, device=DEV)
    ops.conv3x3_wgrad(dy, x0, x1, dw)
    a, b, c = _dot(y, dy), _dot(x, dx), _dot(w, dw)
    scale = float(((y.double() * dy.double()) ** 2).sum().sqrt())     # std of the (cancelling) sum
    for u, v in ((a, b), (a, c), (b, c)):
        assert abs(u - v) <= 5e-3 * (abs(u) + abs(v)) / 2 + 5e-2 * scale, (a, b, c, scale)


def test_conv_adjoint_identity_fp32_split():
    from stain2stain_amd import ops
    B, H, cin, cout, dt = 4, 64, 96, 160, torch.float32
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.rand(B, H, H, cin, device=DEV, generator=g) * 2 - 1
    dy = torch.rand(B, H, H, cout, device=DEV, generator=g) * 2 - 1
    w = (torch.rand(cout, cin, 3, 3, device=DEV, generator=g) * 2 - 1) * 0.05
    wf, wd = ops.pack_conv3x3(w, dt)
    y, _ = ops.conv3x3(x, None, wf, None, cout)
    dx, _ = ops.conv3x3(dy, None, wd, None, cin)
    dw = torch.empty(cout, cin, 3, 3, device=DEV)
    ops.conv3x3_wgrad(dy, x, None, dw)
    a, b, c = _dot(y, dy), _dot(x, dx), _dot(w, dw)
    scale = float(((y.double() * dy.double()) ** 2).sum().sqrt())
    for u, v in ((a, b), (a, c), (b, c)):
        assert abs(u - v) <= 2e-5 * (abs(u) + abs(v)) / 2 + 1e-4 * scale, (a, b, c, scale)


def test_batchnorm_invariants_full_size_forward():
    """Every BatchNorm'd conv output of a 256x256 batch, re-normalised with the forward's own (mean, invstd), has
    mean ~0 and second moment ~1 per channel (bf16 storage of the conv output bounds the deviation)."""
    from stain2stain_amd import FlowUNet, engine, ops
    torch.manual_seed(1984)
    net = FlowUNet().to(DEV).train()
    x = torch.rand(4, 3, 256, 256, device=DEV) * 2 - 1
    ectx = engine.encoder_forward(net.encoder._blocks, x, torch.bfloat16, True)
    for lvl, pair in enumerate(ectx.layers):
        for lc in pair:
            raw = lc.raw.float()
            xh = (raw - lc.stats[0]) * lc.stats[1]
            m = xh.mean((0, 1, 2)).abs().max()
            v = ((xh ** 2).mean((0, 1, 2)) - 1).abs().max()
            assert float(m) < 2e-2 and float(v) < 2e-2, (lvl, float(m), float(v))
            act = lc.act.float()
            assert float(act.min()) >= 0.0
            # ReLU mask = sign of the normalised value: the kernel evaluates fma(raw, scale, shift) in fp32, torch's
            # unfused multiply-add here may round differently, so a disagreement is only legal on a knife edge
            z = raw * lc.stats[2] + lc.stats[3]
            mism = (act > 0) != (z > 0)
            assert float(mism.float().mean()) < 1e-5, (lvl, float(mism.float().mean()))
            assert not bool(mism.any()) or float(z[mism].abs().max()) < 1e-5, (lvl, float(z[mism].abs().max()))


def test_cfg5_tile_size_trains():
    """512x512 tiles (BASELINE.json config 5 geometry, reduced batch): the whole step runs and learns."""
    from stain2stain_amd import CFMTrainer, FlowUNet
    torch.manual_seed(0)
    net = FlowUNet().to(DEV).train()
    tr = CFMTrainer(net, lr=1e-3, weight_decay=0.0)
    x0 = torch.rand(2, 3, 512, 512, device=DEV) * 2 - 1
    x1 = torch.rand(2, 3, 512, 512, device=DEV) * 2 - 1
    t = torch.rand(2, device=DEV)
    losses = [float(tr.step(x0, x1, t)) for _ in range(3)]
    assert all(l == l for l in losses) and losses[-1] < losses[0]
