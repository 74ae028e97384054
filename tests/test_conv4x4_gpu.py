"""4x4 stride-2 convolution (forward, data and weight gradient) and transposed convolution on the 2x2-tap MFMA kernels (the
reference has no such layer, so torch's own operators on bf16-rounded operands are the yardstick).  bf16-stored
outputs: 4e-3 max-norm relative, as for the 3x3 kernel."""
import pytest
import torch
import torch.nn.functional as F

from conftest import relerr

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().to(DEV, BF)


def nchw(y):
    return y.float().cpu().permute(0, 3, 1, 2).contiguous()


def rb(a):
    return a.to(BF).float()


def test_space_to_depth_round_trip():
    from stain2stain_amd import ops, pix2pix as P
    for shape in [(2, 6, 8, 8), (1, 2, 2, 16), (3, 14, 4, 40)]:
        B, H, W, C = shape
        x = (torch.arange(B * H * W * C, dtype=torch.float32, device=DEV).view(shape) % 251 - 125).to(BF)
        ref = P.space_to_depth_pad1_torch(x)
        xs = ops.space_to_depth_pad1(x)
        assert xs.shape == (B, H // 2 + 1, W // 2 + 1, 4 * C) and torch.equal(xs, ref)
        assert torch.equal(ops.depth_to_space_unpad1(xs), x) and torch.equal(P.depth_to_space_unpad1_torch(ref), x)
    x = (torch.arange(2 * 6 * 8 * 8, dtype=torch.float32, device=DEV).view(2, 6, 8, 8) % 251).to(BF)
    xs = ops.space_to_depth_pad1(x)
    assert torch.equal(xs[:, 1, 1, 3 * 8:], x[:, 2, 2, :])          # (r,s) = (1,1) of cell (1,1) is xpad[3,3] = x[2,2]
    assert float(xs[:, 0, :, :16].abs().max()) == 0.0                 # r = 0 of the first cell row is the zero border
    # a channel slice of a wider buffer as the source (pixel stride != C)
    wide = torch.zeros(2, 6, 8, 24, device=DEV, dtype=BF); wide[..., 8:16] = x
    assert torch.equal(ops.space_to_depth_pad1(wide[..., 8:16]), xs)


CASES = [  # B, H, W, Cin, Cout
    (2, 16, 16, 8, 64),        # image layer (RGB padded to 8 channels)
    (2, 32, 64, 64, 128),      # wide map, 128-channel tile
    (1, 20, 12, 24, 40),       # ragged: K = 96 (3 chunks), Cout tail, W tail in the narrow tile
    (3, 8, 8, 128, 256),
    (2, 2, 2, 64, 64),         # innermost levels: 2x2 -> 1x1
    (1, 66, 36, 16, 72),
]


@pytest.mark.parametrize("case", CASES)
def test_conv4x4_stride2_forward_and_input_gradient(case):
    from stain2stain_amd import pix2pix as P
    B, H, W, cin, cout = case
    g = torch.Generator().manual_seed(40 + cin)
    x = rb(torch.rand(B, cin, H, W, generator=g) * 2 - 1).requires_grad_(True)
    w = rb((torch.rand(cout, cin, 4, 4, generator=g) * 2 - 1) * (3.0 / (16 * cin)) ** 0.5)
    b = torch.rand(cout, generator=g) - 0.5
    dy = rb(torch.rand(B, cout, H // 2, W // 2, generator=g) - 0.5)
    ref = F.conv2d(x, w, b, stride=2, padding=1)
    (ref * dy).sum().backward()
    wf, wd = P.pack_conv4x4_s2(w.to(DEV))
    y = P.conv4x4_s2(nhwc(x.detach()), wf, b.to(DEV), cout)
    assert y.shape == (B, H // 2, W // 2, cout)
    assert relerr(nchw(y), ref.detach()) < 4e-3
    dx = P.conv4x4_s2_dgrad(nhwc(dy), wd, cin)
    assert dx.shape == (B, H, W, cin)
    assert relerr(nchw(dx), x.grad) < 4e-3
    wr = w.clone().requires_grad_(True)                            # weight gradient (fp32 output: 1e-4)
    (F.conv2d(x.detach(), wr, None, stride=2, padding=1) * dy).sum().backward()
    gw = P.conv4x4_s2_wgrad(nhwc(dy), nhwc(x.detach()))
    assert gw.shape == (cout, cin, 4, 4)
    assert relerr(gw.cpu(), wr.grad) < 1e-4
    acc = torch.full_like(gw, 2.0)
    P.conv4x4_s2_wgrad(nhwc(dy), nhwc(x.detach()), acc, accumulate=True)
    assert relerr(acc.cpu() - 2.0, wr.grad) < 1e-4


@pytest.mark.parametrize("case", [(2, 8, 8, 128, 64), (1, 1, 1, 64, 64), (2, 5, 7, 40, 24)])
def test_conv_transpose4x4_stride2(case):
    """nn.ConvTranspose2d(k=4, stride=2, padding=1) (the pix2pix decoder's up-convolution) = the data-gradient form."""
    from stain2stain_amd import pix2pix as P
    B, H, W, cin, cout = case                                   # transposed conv: cin -> cout, (H, W) -> (2H, 2W)
    g = torch.Generator().manual_seed(50 + cin)
    x = rb(torch.rand(B, cin, H, W, generator=g) * 2 - 1)
    wt = rb((torch.rand(cin, cout, 4, 4, generator=g) * 2 - 1) * (3.0 / (4 * cin)) ** 0.5)   # ConvTranspose2d layout
    b = torch.rand(cout, generator=g) - 0.5
    ref = F.conv_transpose2d(x, wt, b, stride=2, padding=1)
    _, wd = P.pack_conv4x4_s2(wt.to(DEV))                        # as the Conv2d weight [O = cin, C = cout, 4, 4]
    y = P.conv4x4_s2_dgrad(nhwc(x), wd, cout, bias=b.to(DEV))
    assert y.shape == (B, 2 * H, 2 * W, cout)
    assert relerr(nchw(y), ref) < 4e-3


def test_encoder_level_size_against_torch_on_device():
    """A pix2pix encoder level at the headline batch (16 x 64 channels at 128x128 -> 128 channels at 64x64): forward on
    two samples and the input gradient on two samples against torch's CPU convolution."""
    from stain2stain_amd import pix2pix as P
    g = torch.Generator(device=DEV).manual_seed(1984)
    xs = (torch.rand(16, 128, 128, 64, device=DEV, generator=g) * 2 - 1).to(BF)
    dys = (torch.rand(16, 64, 64, 128, device=DEV, generator=g) - 0.5).to(BF)
    w = ((torch.rand(128, 64, 4, 4, device=DEV, generator=g) * 2 - 1) * (3.0 / 1024) ** 0.5).to(BF).float()
    b = torch.rand(128, device=DEV, generator=g) - 0.5
    wf, wd = P.pack_conv4x4_s2(w)
    y = P.conv4x4_s2(xs, wf, b, 128)
    dx = P.conv4x4_s2_dgrad(dys, wd, 64)
    sel = [0, 15]
    xc = xs[sel].float().cpu().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    ref = F.conv2d(xc, w.cpu(), b.cpu(), stride=2, padding=1)
    (ref * dys[sel].float().cpu().permute(0, 3, 1, 2)).sum().backward()
    assert relerr(nchw(y[sel]), ref.detach()) < 4e-3
    assert relerr(nchw(dx[sel]), xc.grad) < 4e-3
    gw = P.conv4x4_s2_wgrad(dys, xs)                               # sums over the whole batch
    ref_gw = torch.nn.grad.conv2d_weight(xs.float().cpu().permute(0, 3, 1, 2).contiguous(), (128, 64, 4, 4),
                                         dys.float().cpu().permute(0, 3, 1, 2).contiguous(), stride=2, padding=1)
    assert relerr(gw.cpu(), ref_gw) < 1e-4


@pytest.mark.parametrize("kind", ["conv", "transposed", "stride1"])
def test_modules_match_torch_layers_through_autograd(kind):
    """Conv4x4Stride2 / ConvTranspose4x4Stride2 against nn.Conv2d / nn.ConvTranspose2d (k=4, s=2, p=1) with the same
    parameters on bf16-rounded inputs: output and input gradient 4e-3 (bf16-stored), parameter gradients 1e-3 (fp32
    sums of bf16 operands; the upstream gradient is rounded to bf16 on the way in)."""
    from stain2stain_amd.pix2pix import Conv4x4Stride1, Conv4x4Stride2, ConvTranspose4x4Stride2
    g = torch.Generator().manual_seed(77)
    if kind == "stride1":
        mine, ref = Conv4x4Stride1(32, 8).to(DEV), torch.nn.Conv2d(32, 8, 4, 1, 1)
        x = rb(torch.rand(2, 32, 11, 14, generator=g) * 2 - 1)
    elif kind == "conv":
        mine, ref = Conv4x4Stride2(16, 40).to(DEV), torch.nn.Conv2d(16, 40, 4, 2, 1)
        x = rb(torch.rand(2, 16, 12, 20, generator=g) * 2 - 1)
    else:
        mine, ref = ConvTranspose4x4Stride2(24, 16).to(DEV), torch.nn.ConvTranspose2d(24, 16, 4, 2, 1)
        x = rb(torch.rand(2, 24, 6, 10, generator=g) * 2 - 1)
    assert set(mine.state_dict()) == set(ref.state_dict())
    assert all(mine.state_dict()[k].shape == v.shape for k, v in ref.state_dict().items())
    with torch.no_grad():
        for k, v in ref.state_dict().items():
            v.copy_(rb(mine.state_dict()[k].cpu()))
        mine.load_state_dict(ref.state_dict())
    xr = x.clone().requires_grad_(True)
    out_ref = ref(xr)
    cot = rb(torch.rand(out_ref.shape, generator=g) - 0.5)
    (out_ref * cot).sum().backward()
    xg = x.to(DEV).requires_grad_(True)
    out = mine(xg)
    (out.float() * cot.to(DEV)).sum().backward()
    assert relerr(out.float().cpu(), out_ref.detach()) < 4e-3
    assert relerr(xg.grad.cpu(), xr.grad) < 4e-3
    assert relerr(mine.weight.grad.cpu(), ref.weight.grad) < 1e-3
    assert relerr(mine.bias.grad.cpu(), ref.bias.grad) < 1e-3
    # the packed operands follow the master weights
    with torch.no_grad():
        mine.weight.mul_(0.5); ref.weight.mul_(0.5)
    assert relerr(mine(xg).float().cpu(), ref(xr).detach()) < 4e-3


@pytest.mark.parametrize("case", [(2, 17, 33, 64, 128), (1, 9, 12, 24, 8), (2, 32, 32, 256, 512), (3, 5, 40, 40, 72)])
def test_conv4x4_stride1_forward_and_input_gradient(case):
    """nn.Conv2d(k=4, stride=1, padding=1) (the PatchGAN's last two layers; Cout = 1 is padded to 8) on the 16-tap form
    of the loop, and its data gradient."""
    from stain2stain_amd import pix2pix as P
    B, H, W, cin, cout = case
    g = torch.Generator().manual_seed(60 + cin)
    x = rb(torch.rand(B, cin, H, W, generator=g) * 2 - 1).requires_grad_(True)
    w = rb((torch.rand(cout, cin, 4, 4, generator=g) * 2 - 1) * (3.0 / (16 * cin)) ** 0.5)
    b = torch.rand(cout, generator=g) - 0.5
    ref = F.conv2d(x, w, b, stride=1, padding=1)
    dy = rb(torch.rand(ref.shape, generator=g) - 0.5)
    (ref * dy).sum().backward()
    wf, wd = P.pack_conv4x4_s1(w.to(DEV))
    y = P.conv4x4_s1(nhwc(x.detach()), wf, b.to(DEV), cout)
    assert y.shape == (B, H - 1, W - 1, cout)
    assert relerr(nchw(y), ref.detach()) < 4e-3
    dx = P.conv4x4_s1_dgrad(nhwc(dy), wd, cin)
    assert dx.shape == (B, H, W, cin)
    assert relerr(nchw(dx), x.grad) < 4e-3
    wr = w.clone().requires_grad_(True)
    (F.conv2d(x.detach(), wr, None, stride=1, padding=1) * dy).sum().backward()
    gw = P.conv4x4_s1_wgrad(nhwc(dy), nhwc(x.detach()))
    assert gw.shape == (cout, cin, 4, 4)
    assert relerr(gw.cpu(), wr.grad) < 1e-4


@pytest.mark.parametrize("shape", [(64, 8), (40, 24), (136, 72), (8, 512)])
def test_weight_packing_kernel_equals_the_torch_restatement(shape):
    from stain2stain_amd import ops, pix2pix as P
    cout, cin = shape
    w = torch.randn(cout, cin, 4, 4, generator=torch.Generator().manual_seed(cout + cin)).to(DEV)
    for stride, ref in ((2, P.pack_conv4x4_s2_torch), (1, P.pack_conv4x4_s1_torch)):
        wf, wd = ops.pack_conv4x4(w, stride)
        rf, rd = ref(w)
        assert wf.shape == rf.shape and wd.shape == rd.shape
        assert torch.equal(wf, rf) and torch.equal(wd, rd)


@pytest.mark.parametrize("shape", [(2, 5, 7, 24), (16, 64, 64, 128), (3, 1, 1, 8)])
def test_channel_sum(shape):
    from stain2stain_amd import ops
    x = (torch.randn(*shape, generator=torch.Generator().manual_seed(5)) * 2).to(DEV, BF)
    wide = torch.zeros(*shape[:3], shape[3] + 8, device=DEV, dtype=BF); wide[..., 8:] = x
    ref = x.double().sum((0, 1, 2)).float()
    assert relerr(ops.channel_sum(x), ref) < 1e-5
    if shape[1] * shape[2] > 1:                    # a pixel stride is only expressible with more than one pixel per sample
        assert relerr(ops.channel_sum(wide[..., 8:]), ref) < 1e-5
