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


# ------------------------------------------------------------------------------------------------------------------
# 4x4 stride-2 convolution / transposed convolution on the 2x2-tap MFMA kernels (forward, data and weight
# gradients).  The space-to-depth of the padded image and its inverse are one-pass HIP kernels
# (ops.space_to_depth_pad1 / depth_to_space_unpad1; they belong in the apply pass of the producing norm kernel
# eventually), and so is the weight packing (ops.pack_conv4x4).
# ------------------------------------------------------------------------------------------------------------------
def space_to_depth_pad1_torch(x: torch.Tensor) -> torch.Tensor:
    """(torch restatement of ops.space_to_depth_pad1, kept for the tests) NHWC [B,H,W,C] (H, W even) -> [B,H/2+1,W/2+1,4C]: out[p,q,(r*2+s)*C+c] = xpad[2p+r, 2q+s, c], xpad = x with a
    one-pixel zero border."""
    B, H, W, C = x.shape
    if H % 2 or W % 2:
        raise ValueError("space_to_depth_pad1: even spatial size expected")
    xp = torch.nn.functional.pad(x, (0, 0, 1, 1, 1, 1))
    xp = xp.view(B, H // 2 + 1, 2, W // 2 + 1, 2, C).permute(0, 1, 3, 2, 4, 5)
    return xp.reshape(B, H // 2 + 1, W // 2 + 1, 4 * C)


def depth_to_space_unpad1_torch(xs: torch.Tensor) -> torch.Tensor:
    """Inverse of space_to_depth_pad1_torch (drops the border)."""
    B, Hs, Ws, C4 = xs.shape
    C = C4 // 4
    x = xs.view(B, Hs, Ws, 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(B, 2 * Hs, 2 * Ws, C)
    return x[:, 1:-1, 1:-1, :].contiguous()


def _chunks32(m: torch.Tensor) -> torch.Tensor:
    """[rows, taps, K] fp32 -> bf16 [ceil(K/32)][taps][rows][32] (the slab order of the kernel's weight ring)."""
    rows, taps, K = m.shape
    kp = (K + 31) // 32 * 32
    if kp != K:
        m = torch.nn.functional.pad(m, (0, kp - K))
    return m.view(rows, taps, kp // 32, 32).permute(2, 1, 0, 3).contiguous().to(torch.bfloat16)


def pack_conv4x4_s2_torch(w: torch.Tensor):
    """(torch restatement of ops.pack_conv4x4(w, 2), kept for the tests) nn.Conv2d(k=4, s=2, p=1) weight [Cout,Cin,4,4] -> (forward operand, data-gradient operand) of s2s_conv2x2_nhwc.
    Forward: tap (a,b), k = (r*2+s)*Cin + c  <-  w[o, c, 2a+r, 2b+s].  Data gradient: the same with the taps flipped and
    the roles of k and o exchanged."""
    Cout, Cin = w.shape[:2]
    w6 = w.detach().float().view(Cout, Cin, 2, 2, 2, 2)                 # o, c, a, r, b, s
    w2 = w6.permute(0, 2, 4, 3, 5, 1).reshape(Cout, 4, 4 * Cin)         # o, (a,b), (r,s,c)
    fwd = _chunks32(w2)
    wd = w2.flip(1).permute(2, 1, 0).contiguous()                        # k, (1-a,1-b), o
    return fwd, _chunks32(wd)


def pack_conv4x4_s2(w: torch.Tensor):
    """nn.Conv2d(k=4, s=2, p=1) weight [Cout,Cin,4,4] -> (forward operand, data-gradient operand) of s2s_conv2x2_nhwc."""
    return ops.pack_conv4x4(w, 2)


def pack_conv4x4_s1(w: torch.Tensor):
    """nn.Conv2d(k=4, s=1, p=1) weight [Cout,Cin,4,4] -> (forward operand, data-gradient operand) of s2s_conv4x4s1_nhwc."""
    return ops.pack_conv4x4(w, 1)


def conv4x4_s2(x: torch.Tensor, w_fwd: torch.Tensor, bias, cout: int) -> torch.Tensor:
    """nn.Conv2d(k=4, stride=2, padding=1) on an NHWC bf16 tensor [B,H,W,Cin] -> [B,H/2,W/2,cout]."""
    if x.shape[1] % 2 or x.shape[2] % 2:
        raise ValueError("conv4x4_s2: even spatial size expected")
    return ops.conv2x2(ops.space_to_depth_pad1(x), w_fwd, bias, cout, 0)


def conv4x4_s2_dgrad(dy: torch.Tensor, w_dgrad: torch.Tensor, cin: int, bias=None) -> torch.Tensor:
    """Input gradient of conv4x4_s2 (= nn.ConvTranspose2d(k=4, stride=2, padding=1) with weight [Cout,Cin,4,4]):
    NHWC [B,H/2,W/2,Cout] -> [B,H,W,cin].  ``bias`` ([cin]) is the transposed convolution's bias: replicated over the
    four sub-pixel positions it is added by the kernel's epilogue."""
    b4 = None if bias is None else bias.float().repeat(4).contiguous()
    return ops.depth_to_space_unpad1(ops.conv2x2(dy, w_dgrad, b4, 4 * cin, 1))


def conv4x4_s2_wgrad(dy: torch.Tensor, x: torch.Tensor, grad: torch.Tensor = None, accumulate: bool = False) -> torch.Tensor:
    """Weight gradient of conv4x4_s2 in nn.Conv2d's layout [Cout,Cin,4,4] (fp32): dy [B,H/2,W/2,Cout], x [B,H,W,Cin]
    NHWC bf16.  The kernel produces it per 2x2 tap over the space-to-depth channels; the re-ordering into OIHW is a
    view permutation of that small tensor."""
    cout, cin = dy.shape[3], x.shape[3]
    g2 = torch.empty((4, cout, 4 * cin), dtype=torch.float32, device=dy.device)
    ops.conv2x2_wgrad(dy, ops.space_to_depth_pad1(x), g2)
    # g2[(a,b)][o][(r,s,c)] -> w[o][c][2a+r][2b+s]
    g = g2.view(2, 2, cout, 2, 2, cin).permute(2, 5, 0, 3, 1, 4).reshape(cout, cin, 4, 4)
    if grad is None:
        return g.contiguous()
    if accumulate:
        grad.add_(g)
    else:
        grad.copy_(g)
    return grad



def pack_conv4x4_s1_torch(w: torch.Tensor):
    """(torch restatement of ops.pack_conv4x4(w, 1), kept for the tests) nn.Conv2d(k=4, s=1, p=1) weight [Cout,Cin,4,4] -> (forward operand, data-gradient operand) of s2s_conv4x4s1_nhwc:
    [chunk][tap kh*4+kw][rows][32] with rows = Cout, k = Cin forward and rows = Cin, k = Cout, taps flipped backward."""
    Cout, Cin = w.shape[:2]
    wf = w.detach().float().permute(0, 2, 3, 1).reshape(Cout, 16, Cin)                    # o, (kh,kw), c
    wd = w.detach().float().flip(2, 3).permute(1, 2, 3, 0).reshape(Cin, 16, Cout)         # c, (3-kh,3-kw), o
    return _chunks32(wf), _chunks32(wd)


def conv4x4_s1(x: torch.Tensor, w_fwd: torch.Tensor, bias, cout: int) -> torch.Tensor:
    """nn.Conv2d(k=4, stride=1, padding=1) on an NHWC bf16 tensor [B,H,W,Cin] -> [B,H-1,W-1,cout]."""
    return ops.conv4x4s1(x, w_fwd, bias, cout, 1)


def conv4x4_s1_dgrad(dy: torch.Tensor, w_dgrad: torch.Tensor, cin: int) -> torch.Tensor:
    """Input gradient of conv4x4_s1: [B,H-1,W-1,Cout] -> [B,H,W,cin]."""
    return ops.conv4x4s1(dy, w_dgrad, None, cin, 2)


def conv4x4_s1_wgrad(dy: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """Weight gradient of conv4x4_s1 in nn.Conv2d's layout [Cout,Cin,4,4] (fp32): dy [B,H-1,W-1,Cout], x [B,H,W,Cin]."""
    cout, cin = dy.shape[3], x.shape[3]
    g16 = torch.empty((16, cout, cin), dtype=torch.float32, device=dy.device)
    ops.conv4x4s1_wgrad(dy, x, g16)
    return g16.view(4, 4, cout, cin).permute(2, 3, 0, 1).contiguous()

# ------------------------------------------------------------------------------------------------------------------
# nn.Module faces of the layers (signatures and state-dict keys of nn.Conv2d / nn.ConvTranspose2d with
# kernel_size=4, padding=1, stride 2 or - PatchGAN - 1).  Activations are bf16 NHWC in memory (logical NCHW in torch.channels_last);
# weights stay fp32 masters and are packed into the MFMA operand layouts whenever they change.
# ------------------------------------------------------------------------------------------------------------------
class _PackCache:
    def __init__(self, pack=None):
        self.key, self.packed, self.pack = None, None, pack

    def get(self, w: torch.Tensor):
        key = (w.data_ptr(), w._version, w.device)
        if key != self.key:
            self.packed, self.key = (self.pack or pack_conv4x4_s2)(w), key
        return self.packed


def _to_nhwc_bf16(x: torch.Tensor) -> torch.Tensor:
    if not x.is_cuda:
        raise RuntimeError("stain2stain_amd: the pix2pix layers run on the GPU only (no CPU fallback)")
    return x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)


class _Conv4x4S2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, cache):
        xs = _to_nhwc_bf16(x)
        wf, wd = cache.get(weight)
        y = conv4x4_s2(xs, wf, bias, weight.shape[0])
        ctx.save_for_backward(xs)
        ctx.wd, ctx.cin, ctx.has_bias, ctx.xdtype = wd, weight.shape[1], bias is not None, x.dtype
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        (xs,) = ctx.saved_tensors
        g = _to_nhwc_bf16(gy)
        dx = conv4x4_s2_dgrad(g, ctx.wd, ctx.cin).permute(0, 3, 1, 2).to(ctx.xdtype) if ctx.needs_input_grad[0] else None
        dw = conv4x4_s2_wgrad(g, xs) if ctx.needs_input_grad[1] else None      # frozen D in the generator pass
        db = ops.channel_sum(g) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dw, db, None


class _ConvT4x4S2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, cache):
        xs = _to_nhwc_bf16(x)
        wf, wd = cache.get(weight)                      # weight [Cin, Cout, 4, 4] read as the Conv2d weight [O, C, 4, 4]
        y = conv4x4_s2_dgrad(xs, wd, weight.shape[1], bias=bias)
        ctx.save_for_backward(xs)
        ctx.wf, ctx.cin, ctx.has_bias, ctx.xdtype = wf, weight.shape[0], bias is not None, x.dtype
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        (xs,) = ctx.saved_tensors
        g = _to_nhwc_bf16(gy)                           # [B, 2H, 2W, Cout]: the "input" of the equivalent convolution
        dx = conv4x4_s2(g, ctx.wf, None, ctx.cin).permute(0, 3, 1, 2).to(ctx.xdtype) if ctx.needs_input_grad[0] else None
        dw = conv4x4_s2_wgrad(xs, g) if ctx.needs_input_grad[1] else None    # [O = Cin, C = Cout, 4, 4] = nn.ConvTranspose2d's layout
        db = ops.channel_sum(g) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dw, db, None


class _Layer4x4(nn.Module):
    def __init__(self, w_shape, n_bias, bias: bool, fan_in: int):
        super().__init__()
        if w_shape[0] % 8 or w_shape[1] % 8:
            raise ValueError("channel counts must be multiples of 8 (pad the image to 8 channels)")
        self.weight = nn.Parameter(torch.empty(*w_shape, 4, 4))
        self.bias = nn.Parameter(torch.empty(n_bias)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)                  # nn.Conv2d's default initialisation
        if bias:
            bound = 1.0 / fan_in ** 0.5
            nn.init.uniform_(self.bias, -bound, bound)
        self._cache = _PackCache()


class Conv4x4Stride2(_Layer4x4):
    """``nn.Conv2d(in_channels, out_channels, kernel_size=4, stride=2, padding=1)`` (pix2pix encoder / PatchGAN layer)."""

    def __init__(self, in_channels: int, out_channels: int, bias: bool = True):
        super().__init__((out_channels, in_channels), out_channels, bias, in_channels * 16)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() != 4 or x.shape[1] != self.weight.shape[1] or x.shape[2] % 2 or x.shape[3] % 2:
            raise ValueError(f"expected [B, {self.weight.shape[1]}, even H, even W], got {tuple(x.shape)}")
        return _Conv4x4S2.apply(x, self.weight, self.bias, self._cache)


class ConvTranspose4x4Stride2(_Layer4x4):
    """``nn.ConvTranspose2d(in_channels, out_channels, kernel_size=4, stride=2, padding=1)`` (pix2pix decoder layer)."""

    def __init__(self, in_channels: int, out_channels: int, bias: bool = True):
        super().__init__((in_channels, out_channels), out_channels, bias, out_channels * 16)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() != 4 or x.shape[1] != self.weight.shape[0]:
            raise ValueError(f"expected [B, {self.weight.shape[0]}, H, W], got {tuple(x.shape)}")
        return _ConvT4x4S2.apply(x, self.weight, self.bias, self._cache)


class _Conv4x4S1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, cache):
        xs = _to_nhwc_bf16(x)
        wf, wd = cache.get(weight)
        y = conv4x4_s1(xs, wf, bias, weight.shape[0])
        ctx.save_for_backward(xs)
        ctx.wd, ctx.cin, ctx.has_bias, ctx.xdtype = wd, weight.shape[1], bias is not None, x.dtype
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        (xs,) = ctx.saved_tensors
        g = _to_nhwc_bf16(gy)
        dx = conv4x4_s1_dgrad(g, ctx.wd, ctx.cin).permute(0, 3, 1, 2).to(ctx.xdtype) if ctx.needs_input_grad[0] else None
        dw = conv4x4_s1_wgrad(g, xs) if ctx.needs_input_grad[1] else None
        db = ops.channel_sum(g) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dw, db, None


class Conv4x4Stride1(_Layer4x4):
    """``nn.Conv2d(in_channels, out_channels, kernel_size=4, stride=1, padding=1)`` (the PatchGAN discriminator's last
    two layers; pad the single output logit channel to 8 and read channel 0)."""

    def __init__(self, in_channels: int, out_channels: int, bias: bool = True):
        super().__init__((out_channels, in_channels), out_channels, bias, in_channels * 16)
        self._cache = _PackCache(pack_conv4x4_s1)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() != 4 or x.shape[1] != self.weight.shape[1] or x.shape[2] < 2 or x.shape[3] < 2:
            raise ValueError(f"expected [B, {self.weight.shape[1]}, H >= 2, W >= 2], got {tuple(x.shape)}")
        return _Conv4x4S1.apply(x, self.weight, self.bias, self._cache)


# ------------------------------------------------------------------------------------------------------------------
# The two networks (pix2pix "unet_256"-style generator, 70x70 PatchGAN discriminator).  Under autograd each network is
# ONE node over the passes of the fused engine (pix2pix_engine.NetRunner): image packing, every convolution with its
# activation epilogue, the norm passes that write the ReLU'd skip tensors into the concatenation buffers, tanh -- all HIP
# kernels of this library, none of torch's (no leaky_relu / relu / cat / pad / tanh).  The optimisation step itself is
# ``Pix2PixTrainer.step`` (pix2pix_engine.py); its torch restatement for the parity tests lives with the oracle
# (oracle/pix2pix_oracle.py: pix2pix_losses, pix2pix_step).
# ------------------------------------------------------------------------------------------------------------------
def _runner(net, precision: str):
    from .pix2pix_engine import NetRunner
    r = net.__dict__.get("_runner_obj")
    if r is None or r.dtype != (torch.bfloat16 if precision == "bf16" else torch.float32):
        r = NetRunner(net, precision)
        net.__dict__["_runner_obj"] = r
    r.refresh()
    return r


def _check_versions(net, versions, what: str) -> None:
    """The backward pass reads the runner's PACKED operands, which ``refresh()`` re-packs in place whenever a parameter
    changes: a backward through a forward that ran on older weights (forward, in-place optimiser step, second forward,
    THEN the first output's backward) would silently use the new ones -- raise, as torch's own layers do ("modified by an
    inplace operation")."""
    if tuple(p._version for p in net.parameters()) != versions:
        raise RuntimeError(f"{what}: a parameter needed for the gradient computation has been modified by an inplace "
                           "operation since this forward pass (run backward before the optimiser step)")


def _check_image(x: torch.Tensor, channels: int, what: str) -> torch.Tensor:
    if not x.is_cuda:
        raise RuntimeError("stain2stain_amd: the pix2pix networks run on the GPU only (no CPU fallback)")
    if x.dim() != 4 or x.shape[1] != channels:
        raise ValueError(f"{what}: expected [B, {channels}, H, W], got {tuple(x.shape)}")
    return x.detach().contiguous().float()


class _GeneratorFn(torch.autograd.Function):
    """fake = tanh(U-Net(x)); parameters in ``named_parameters`` order after (net, x)."""

    @staticmethod
    def forward(ctx, net, x, *params):
        r = _runner(net, net.precision)
        src = _check_image(x, net.in_channels, "Pix2PixGenerator")
        g = r.g_forward(src)
        B, _, H, W = src.shape
        C = net.out_channels
        fake = torch.empty((B, C, H, W), dtype=torch.float32, device=src.device)
        ref = src if src.shape[1] == C else fake                 # (the kernel's L1 / D-input outputs are not used here)
        ops.p2p_tanh_l1_fwd(g.h, ref, ref, torch.empty((B, H, W, 8), dtype=r.dtype, device=src.device), fake)
        ctx.runner, ctx.g, ctx.net = r, g, net
        ctx.versions = tuple(p._version for p in net.parameters())
        return fake

    @staticmethod
    def backward(ctx, gfake):
        r, g, net = ctx.runner, ctx.g, ctx.net
        if g is None:
            raise RuntimeError("Pix2PixGenerator: the activations of this forward pass were released by its first backward")
        if ctx.needs_input_grad[1]:
            raise RuntimeError("Pix2PixGenerator: no gradient with respect to the input image")
        _check_versions(net, ctx.versions, "Pix2PixGenerator")
        B, H, W, _ = g.h.shape
        gf = gfake.contiguous().float()
        # d(pre-tanh) = gfake * (1 - fake^2): the tanh backward kernel with the L1 term switched off and the upstream
        # gradient in the channels where it expects the discriminator's input gradient
        gd = ops.p2p_pack_input(gf, gf, torch.empty((B, H, W, 8), dtype=r.dtype, device=gf.device))
        dh = ops.p2p_tanh_l1_bwd(g.h, gf, gd, 0.0)
        r.pG.grads = r.new_grads()
        r.g_backward(g, dh)
        ctx.g = None
        names = [k for k, _ in net.named_parameters()]
        return (None, None) + tuple(r.pG.grads[k] if need else None for k, need in zip(names, ctx.needs_input_grad[2:]))


class _DiscriminatorFn(torch.autograd.Function):
    """logits = PatchGAN(cat(a, b)) as [B, 1, h, w] fp32; gradients for a, b and the parameters."""

    @staticmethod
    def forward(ctx, net, a, b, *params):
        r = _runner(net, net.precision)
        ca = a.shape[1] if a.dim() == 4 else -1
        a32 = _check_image(a, ca, "PatchGANDiscriminator")
        b32 = _check_image(b, net.in_channels - ca, "PatchGANDiscriminator")
        if a32.shape[0] != b32.shape[0] or a32.shape[2:] != b32.shape[2:]:
            raise ValueError("PatchGANDiscriminator: the two images differ in batch or size")
        B, _, H, W = a32.shape
        d_in = ops.p2p_pack_input(a32, b32, torch.empty((B, H, W, 8), dtype=r.dtype, device=a32.device))
        z, saved = r.d_forward(d_in)
        ctx.runner, ctx.saved, ctx.net, ctx.ca, ctx.cb = r, saved, net, ca, net.in_channels - ca
        ctx.versions = tuple(p._version for p in net.parameters())
        return ops.p2p_unpack(z, 0, 1)

    @staticmethod
    def backward(ctx, gz):
        r, net = ctx.runner, ctx.net
        if ctx.saved is None:
            raise RuntimeError("PatchGANDiscriminator: the activations of this forward pass were released by its first backward")
        _check_versions(net, ctx.versions, "PatchGANDiscriminator")
        gz = gz.contiguous().float()
        B, _, h, w = gz.shape
        dz = ops.p2p_pack_input(gz, None, torch.empty((B, h, w, 8), dtype=r.dtype, device=gz.device))
        want_w = any(ctx.needs_input_grad[3:])
        need_x = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        if want_w:
            r.pD.grads = r.new_grads()
        gd = r.d_backward(ctx.saved, dz, want_w=want_w, need_input_grad=need_x)
        ctx.saved = None
        ga = ops.p2p_unpack(gd, 0, ctx.ca) if ctx.needs_input_grad[1] else None
        gb = ops.p2p_unpack(gd, ctx.ca, ctx.cb) if ctx.needs_input_grad[2] else None
        names = [k for k, _ in net.named_parameters()]
        return (None, ga, gb) + tuple(r.pD.grads[k] if need else None for k, need in zip(names, ctx.needs_input_grad[3:]))


class Pix2PixGenerator(nn.Module):
    """U-Net generator: ``num_downs`` 4x4 stride-2 convolutions down to 1x1 (8 for 256x256 tiles) and the mirrored
    transposed convolutions with skip connections; LeakyReLU(0.2) / InstanceNorm on the way down, ReLU / InstanceNorm
    on the way up, tanh at the end.  Channel plan ngf * (1, 2, 4, 8, 8, ...).  ``precision``: "bf16" or "fp32" (the
    three-way-split parity mode) for the module's own ``forward``; ``Pix2PixTrainer`` takes its own."""

    def __init__(self, in_channels: int = 3, out_channels: int = 3, ngf: int = 64, num_downs: int = 8,
                 precision: str = "bf16"):
        super().__init__()
        if num_downs < 2:                 # (2 = BASELINE.json configs[0]'s "2-level U-Net": outermost + innermost layer)
            raise ValueError("num_downs >= 2")
        if precision not in ("bf16", "fp32"):
            raise ValueError("precision must be 'bf16' or 'fp32'")
        ch = [ngf * min(2 ** i, 8) for i in range(num_downs)]
        self.in_channels, self.out_channels, self.precision = in_channels, out_channels, precision
        self.downs = nn.ModuleList([Conv4x4Stride2(8 if i == 0 else ch[i - 1], ch[i]) for i in range(num_downs)])
        self.down_norms = nn.ModuleList([InstanceNormLeakyReLU(ch[i], negative_slope=0.2) for i in range(1, num_downs - 1)])
        ups, norms = [], []
        for i in range(num_downs - 1, -1, -1):                      # innermost first
            cin = ch[i] if i == num_downs - 1 else 2 * ch[i]
            cout = 8 if i == 0 else ch[i - 1]
            ups.append(ConvTranspose4x4Stride2(cin, cout))
            if i > 0:
                norms.append(InstanceNormLeakyReLU(cout, negative_slope=0.0))
        self.ups, self.up_norms = nn.ModuleList(ups), nn.ModuleList(norms)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [B, in_channels, H, W] -> tanh output [B, out_channels, H, W] (fp32); differentiable in the parameters."""
        return _GeneratorFn.apply(self, x, *self.parameters())


class PatchGANDiscriminator(nn.Module):
    """PatchGAN: C(ndf) stride 2 with LeakyReLU(0.2), ``n_layers - 1`` more stride-2 layers and one stride-1 layer with
    InstanceNorm + LeakyReLU (channels doubling up to 8 ndf), then the stride-1 logit layer; input = cat(source, target
    or generated) along the channels.  ``n_layers = 3`` is the 70x70 PatchGAN C64 - C128 - C256 - C512 - C1 (layers
    ``c1`` ... ``c5``, norms ``n2`` ... ``n4``); ``n_layers = 1`` is BASELINE.json configs[0]'s "1-layer PatchGAN"
    (C64 - C128 - C1: ``c1``, ``c2`` / ``n2``, ``c3``)."""

    def __init__(self, in_channels: int = 6, ndf: int = 64, n_layers: int = 3, precision: str = "bf16"):
        super().__init__()
        if n_layers < 1:
            raise ValueError("n_layers >= 1")
        if precision not in ("bf16", "fp32"):
            raise ValueError("precision must be 'bf16' or 'fp32'")
        if not 2 <= in_channels <= 8:
            raise ValueError("in_channels: 2 ... 8 (the two images together, padded to 8 channels)")
        self.n_layers, self.in_channels, self.precision = n_layers, in_channels, precision
        self.c1 = Conv4x4Stride2(8, ndf)
        kinds = ["s2"]
        cin = ndf
        for n in range(1, n_layers + 1):
            cout = ndf * min(2 ** n, 8)
            stride1 = n == n_layers
            setattr(self, f"c{n + 1}", (Conv4x4Stride1 if stride1 else Conv4x4Stride2)(cin, cout))
            setattr(self, f"n{n + 1}", InstanceNormLeakyReLU(cout))
            kinds.append("s1" if stride1 else "s2")
            cin = cout
        setattr(self, f"c{n_layers + 2}", Conv4x4Stride1(cin, 8))      # one logit channel, padded to 8
        kinds.append("s1")
        self.layer_kinds = kinds                                        # kinds[k] of layer c{k+1}

    def conv_layers(self):
        """[(name, kind, module)] in forward order."""
        return [(f"c{k + 1}", kind, getattr(self, f"c{k + 1}")) for k, kind in enumerate(self.layer_kinds)]

    def forward(self, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        """a, b: the two images (NCHW, ``in_channels`` together) -> logits [B, 1, h, w] (fp32); differentiable in a, b and
        the parameters."""
        return _DiscriminatorFn.apply(self, a, b, *self.parameters())
