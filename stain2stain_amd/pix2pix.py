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
# The two networks (pix2pix "unet_256"-style generator, 70x70 PatchGAN discriminator) assembled from the layers above.
# The convolutions and the norm + activation pairs run on the HIP kernels; what is still torch glue here - the
# activation without a norm (first / innermost layer), the ReLU of the skip tensors, the channel concatenations, tanh,
# the losses and the optimiser - is bandwidth-bound elementwise work waiting for its kernels.
# ------------------------------------------------------------------------------------------------------------------
def _pad_channels(x: torch.Tensor, c: int) -> torch.Tensor:
    return x if x.shape[1] == c else torch.nn.functional.pad(x, (0, 0, 0, 0, 0, c - x.shape[1]))


class Pix2PixGenerator(nn.Module):
    """U-Net generator: ``num_downs`` 4x4 stride-2 convolutions down to 1x1 (8 for 256x256 tiles) and the mirrored
    transposed convolutions with skip connections; LeakyReLU(0.2) / InstanceNorm on the way down, ReLU / InstanceNorm
    on the way up, tanh at the end.  Channel plan ngf * (1, 2, 4, 8, 8, ...)."""

    def __init__(self, in_channels: int = 3, out_channels: int = 3, ngf: int = 64, num_downs: int = 8):
        super().__init__()
        if num_downs < 2:                 # (2 = BASELINE.json configs[0]'s "2-level U-Net": outermost + innermost layer)
            raise ValueError("num_downs >= 2")
        ch = [ngf * min(2 ** i, 8) for i in range(num_downs)]
        self.in_channels, self.out_channels = in_channels, out_channels
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
        n = len(self.downs)
        acts = []
        h = self.downs[0](_pad_channels(x, 8))
        h = torch.nn.functional.leaky_relu(h, 0.2)
        acts.append(h)
        for i in range(1, n - 1):
            h = self.down_norms[i - 1](self.downs[i](h))
            acts.append(h)
        h = torch.relu(self.downs[n - 1](h))                       # innermost: no norm
        for j, up in enumerate(self.ups):
            h = up(h)
            if j < n - 1:
                h = self.up_norms[j](h)
                h = torch.cat([torch.relu(acts[n - 2 - j]), h], 1)
        return torch.tanh(h[:, :self.out_channels].float())


class PatchGANDiscriminator(nn.Module):
    """PatchGAN: C(ndf) stride 2 with LeakyReLU(0.2), ``n_layers - 1`` more stride-2 layers and one stride-1 layer with
    InstanceNorm + LeakyReLU (channels doubling up to 8 ndf), then the stride-1 logit layer; input = cat(source, target
    or generated) along the channels.  ``n_layers = 3`` is the 70x70 PatchGAN C64 - C128 - C256 - C512 - C1 (layers
    ``c1`` ... ``c5``, norms ``n2`` ... ``n4``); ``n_layers = 1`` is BASELINE.json configs[0]'s "1-layer PatchGAN"
    (C64 - C128 - C1: ``c1``, ``c2`` / ``n2``, ``c3``)."""

    def __init__(self, in_channels: int = 6, ndf: int = 64, n_layers: int = 3):
        super().__init__()
        if n_layers < 1:
            raise ValueError("n_layers >= 1")
        self.n_layers = n_layers
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
        K = len(self.layer_kinds)
        h = torch.nn.functional.leaky_relu(self.c1(_pad_channels(torch.cat([a, b], 1), 8)), 0.2)
        for k in range(2, K):
            h = getattr(self, f"n{k}")(getattr(self, f"c{k}")(h))
        return getattr(self, f"c{K}")(h)[:, :1].float()


def pix2pix_losses(G: nn.Module, D: nn.Module, src: torch.Tensor, tgt: torch.Tensor, lambda_l1: float = 100.0):
    """One pix2pix evaluation: (fake, loss_D, loss_G) with the vanilla GAN objective (BCE with logits) + lambda * L1."""
    bce = torch.nn.functional.binary_cross_entropy_with_logits
    fake = G(src)
    pr, pf = D(src, tgt), D(src, fake.detach())
    loss_d = 0.5 * (bce(pr, torch.ones_like(pr)) + bce(pf, torch.zeros_like(pf)))
    pg = D(src, fake)
    loss_g = bce(pg, torch.ones_like(pg)) + lambda_l1 * (fake - tgt).abs().mean()
    return fake, loss_d, loss_g


def pix2pix_step(G, D, opt_g, opt_d, src, tgt, lambda_l1: float = 100.0):
    """The G + D optimisation step of pix2pix: discriminator update on (real, detached fake), then generator update
    through the discriminator.  Returns (loss_D, loss_G) as tensors."""
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
