"""The pix2pix G + D optimisation step without autograd and without torch math.

BASELINE.json's north_star opens with this path ("pix2pix-style U-Net generator + PatchGAN discriminator forward /
backward ... fused InstanceNorm+LeakyReLU") and its ``metric`` is worded on it ("G+D step"); the reference repository
holds no such model, so the semantics are those of ``oracle/pix2pix_oracle.py`` -- the same two networks
built from torch's own layers -- and parity with the reference is unpinned by construction.

One step (``Pix2PixTrainer.step``), every launch a HIP kernel of libstain2stain_hip.so:

    fake            <- G(src)                                   (generator forward, tanh head)
    D update        <- BCE(D(src, tgt), 1) / 2 + BCE(D(src, fake.detach()), 0) / 2
                       real and fake ride through D as ONE batch of 2B samples (InstanceNorm is per sample, so the
                       result equals two passes); discriminator backward, [all-reduce], fused Adam, repack
    G update        <- BCE(D(src, fake), 1) + lambda * L1(fake, tgt) through the UPDATED D (data gradient only: no
                       discriminator weight gradients are formed), generator backward, [all-reduce], fused Adam, repack

Layout: NHWC, bf16 (throughput) or fp32 (three-way-split parity mode); images padded to 8 channels.  The 4x4 stride-2
convolution is a 2x2 convolution over the space-to-depth image of its padded input, the transposed convolution the same
loop with flipped taps (conv3x3_mfma.hip, convkxk).  The skip connections never materialise ``torch.cat``: the encoder's
norm pass writes relu(z) into the first half of the decoder's concatenation buffer and the decoder's norm pass writes its
output into the second half.  Parameters, gradients and Adam moments of each network live in flat fp32 buffers in
backward-completion order (``trainer.FlatParams``), so gradient buckets are contiguous slices.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from . import engine, ops
from .ddp import ALIGN, GradBucketer, all_reduce_mean_scalar, broadcast_from_rank0
from .pix2pix import PatchGANDiscriminator, Pix2PixGenerator

LRELU = 0.2
# The inner U-Net levels' split-K convolutions leave their reduce to the single-launch InstanceNorm that consumes them
# (ops.SplitSum; 11 launches fewer per step, the same bits).  S2S_P2P_FOLD_IN_NORM=0: separate reduce launches.
FOLD_IN_NORM = os.environ.get("S2S_P2P_FOLD_IN_NORM", "1") != "0"


def _check_norms(net) -> None:
    """The fused passes hard-wire InstanceNorm2d(affine=False, eps=1e-5), the form both networks are built with: a norm
    module edited to anything else must not be silently ignored."""
    from .pix2pix import InstanceNormLeakyReLU
    for name, m in net.named_modules():
        if isinstance(m, InstanceNormLeakyReLU) and (abs(m.eps - 1e-5) > 1e-12 or m.weight is not None):
            raise NotImplementedError(f"stain2stain_amd: {name}: the fused pix2pix passes implement "
                                      "InstanceNorm2d(affine=False, eps=1e-5) only")


class FlatParams:
    """Parameters of one network as views of a flat fp32 buffer + flat gradient / Adam-moment buffers.

    ``groups``: [[(name, parameter), ...], ...] in the order the backward pass completes them."""

    def __init__(self, groups: List[List[Tuple[str, torch.nn.Parameter]]], bucket_mb: float, process_group,
                 max_bucket_mb: float = 16.0, sharded: bool = False):
        dev = groups[0][0][1].device
        sizes, offs, off = [], {}, 0
        self.group_range: List[Tuple[int, int]] = []
        for g in groups:
            start = off
            for name, p in g:
                offs[name] = off
                off += (p.numel() + 7) // 8 * 8            # 32-byte aligned views
            off = (off + ALIGN - 1) // ALIGN * ALIGN       # groups (layers) divide into aligned shards for <= 8 ranks
            sizes.append(off - start)
            self.group_range.append((start, off))
        self.p = torch.zeros(off, dtype=torch.float32, device=dev)
        self.g = torch.zeros(off, dtype=torch.float32, device=dev)
        self.m = torch.zeros(off, dtype=torch.float32, device=dev)
        self.v = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grads: Dict[str, torch.Tensor] = {}
        self.slot: Dict[str, Tuple[int, int]] = {}
        with torch.no_grad():
            for g in groups:
                for name, p in g:
                    o, n = offs[name], p.numel()
                    self.p[o:o + n].copy_(p.detach().reshape(-1))
                    p.data = self.p[o:o + n].view(p.shape)
                    p.grad = self.g[o:o + n].view(p.shape)
                    self.grads[name] = p.grad
                    self.slot[name] = (o, n)
        self.bucketer = GradBucketer(self.g, sizes, bucket_mb, process_group, max_bucket_mb,
                                     "reduce_scatter" if sharded else "allreduce")
        self.step_count = 0
        broadcast_from_rank0([self.p], process_group)

    def adam(self, lr: float, betas: Tuple[float, float], eps: float, weight_decay: float,
             hyper_dev: Optional[torch.Tensor] = None) -> None:
        """``hyper_dev``: the Adam scalars in device memory (a captured step; the caller advances ``step_count`` and
        refreshes them before each replay) instead of kernel arguments."""
        self.bucketer.wait_all()
        if hyper_dev is None:
            self.step_count += 1
        for lo, hi in self.bucketer.shards():              # everything, or this rank's slices in sharded mode
            if hyper_dev is None:
                ops.adam_step_(self.p[lo:hi], self.g[lo:hi], self.m[lo:hi], self.v[lo:hi], self.step_count, lr, betas[0],
                               betas[1], eps, weight_decay, self.bucketer.grad_scale)
            else:
                ops.adam_step_dev_(self.p[lo:hi], self.g[lo:hi], self.m[lo:hi], self.v[lo:hi], hyper_dev)
        self.bucketer.all_gather(self.p)

    def adam_group(self, group: int, lr: float, betas: Tuple[float, float], eps: float, weight_decay: float,
                   hyper_dev: Optional[torch.Tensor] = None) -> None:
        """Adam on the slice of parameter group ``group`` alone (the step counter is the caller's: ``begin_step``), for
        the update-in-backward of a trainer without a gradient exchange.  Adam is element-wise: the slices of all groups
        together are exactly ``adam``."""
        lo, hi = self.group_range[group]
        if hyper_dev is None:
            ops.adam_step_(self.p[lo:hi], self.g[lo:hi], self.m[lo:hi], self.v[lo:hi], self.step_count, lr, betas[0],
                           betas[1], eps, weight_decay, self.bucketer.grad_scale)
        else:
            ops.adam_step_dev_(self.p[lo:hi], self.g[lo:hi], self.m[lo:hi], self.v[lo:hi], hyper_dev)

    def state_dict(self) -> Dict:
        """Adam state by parameter name.  Sharded optimiser: the moments are all-gathered first (a collective: every rank
        calls), so the dump is complete on every rank."""
        if self.bucketer.mode == "reduce_scatter" and self.bucketer.enabled and self.step_count > 0:
            self.bucketer.all_gather(self.m)
            self.bucketer.all_gather(self.v)
        out = {"step": self.step_count, "exp_avg": {}, "exp_avg_sq": {}}
        for name, (o, n) in self.slot.items():
            shape = self.grads[name].shape
            out["exp_avg"][name] = self.m[o:o + n].view(shape).clone()
            out["exp_avg_sq"][name] = self.v[o:o + n].view(shape).clone()
        return out

    def load_state_dict(self, sd: Dict) -> None:
        if set(sd["exp_avg"]) != set(self.slot) or set(sd["exp_avg_sq"]) != set(self.slot):
            raise ValueError("optimizer state does not match this network's parameters")
        for name, (o, n) in self.slot.items():
            for buf, key in ((self.m, "exp_avg"), (self.v, "exp_avg_sq")):
                t = sd[key][name]
                if tuple(t.shape) != tuple(self.grads[name].shape):
                    raise ValueError(f"optimizer state {name}: shape {tuple(t.shape)} != {tuple(self.grads[name].shape)}")
                buf[o:o + n].copy_(t.reshape(-1))
        self.step_count = int(sd["step"])


@dataclass
class _Layer:
    """One 4x4 layer: fp32 master weight / bias (views of the flat buffer) + its two packed MFMA operands."""
    name: str
    kind: str                           # "s2" conv stride 2, "t2" transposed conv stride 2, "s1" conv stride 1
    weight: torch.nn.Parameter
    bias: Optional[torch.nn.Parameter]
    wf: torch.Tensor = None
    wd: torch.Tensor = None

    @property
    def conv_out(self) -> int:          # output channels of the layer
        return self.weight.shape[1] if self.kind == "t2" else self.weight.shape[0]

    @property
    def conv_in(self) -> int:
        return self.weight.shape[0] if self.kind == "t2" else self.weight.shape[1]


class _Packer:
    """(Re)packs every layer of a network in one launch (s2s_pack_conv4x4_batched)."""

    def __init__(self, layers: List[_Layer], dtype: torch.dtype):
        self.layers, self.dtype = layers, dtype
        rows, start = [], 0
        L = ops._L()
        for l in layers:
            w = l.weight
            o, c = w.shape[0], w.shape[1]                      # nn.ConvTranspose2d's [Cin, Cout] is read as Conv2d's [O, C]
            stride = 1 if l.kind == "s1" else 2
            taps, K = (4, 4 * c) if stride == 2 else (16, c)
            l.wf = torch.empty(((K + 31) // 32, taps, o, 32), dtype=dtype, device=w.device)
            l.wd = torch.empty(((o + 31) // 32, taps, K, 32), dtype=dtype, device=w.device)
            rows.append([w.data_ptr(), l.wf.data_ptr(), l.wd.data_ptr(), o, c, 1 if stride == 2 else 0, start])
            start += L.s2s_pack_conv4x4_blocks(o, c, stride)
        dev = layers[0].weight.device
        self.desc = torch.tensor(rows, dtype=torch.int64, device=dev)
        self.total = start
        # one-layer descriptors (first block 0) for the update-in-backward
        firsts = [r[6] for r in rows] + [start]
        self.one = {l.name: (torch.tensor([r[:6] + [0]], dtype=torch.int64, device=dev), firsts[i + 1] - firsts[i])
                    for i, (l, r) in enumerate(zip(layers, rows))}
        self.repack()

    def repack(self) -> None:
        ops.pack_conv4x4_batched(self.desc, self.total, self.dtype)

    def repack_layer(self, name: str) -> None:
        desc, total = self.one[name]
        ops.pack_conv4x4_batched(desc, total, self.dtype)


# ----------------------------------------------------------------------------------------------------------------------
# layer forward / backward on the kernels
# ----------------------------------------------------------------------------------------------------------------------
def _b(l: "_Layer"):
    return None if l.bias is None else l.bias.detach()


def _conv_s2_fwd(l: _Layer, x: torch.Tensor, act: bool = False, slope: float = 0.0, out2=None, defer: bool = False):
    """nn.Conv2d(k=4, s=2, p=1) of the plain NHWC tensor x.  Returns (y, saved): ``saved`` is what the weight gradient
    reads -- x itself on the layout-free bf16 kernels (the loader does the space-to-depth in its addresses), else the
    explicit space-to-depth image (fp32 parity mode)."""
    bias = None if l.bias is None else l.bias.detach()
    if ops.fused_s2_ok(x.dtype, x.shape[3]):      # (defer: a split launch leaves its reduce to the norm that follows)
        return ops.conv4x4s2(x, l.wf, bias, l.conv_out, act=act, slope=slope, out2=out2, defer=defer), x
    xs = ops.space_to_depth_pad1_t(x)
    return ops.convkxk(xs, l.wf, bias, l.conv_out, 2, 0, act=act, slope=slope, out2=out2), xs


def _conv_s2_bwd(l: _Layer, g: torch.Tensor, saved: torch.Tensor, gw: torch.Tensor, need_dx: bool, want_w: bool = True,
                 defer: bool = False):
    if want_w:      # (a leaf of the backward chain: on the side stream when the trainer has one, engine.run_on_side)
        engine.run_on_side(lambda: ops.convkxk_wgrad(g, saved, gw, 2, x_plain=(saved.shape[3] == l.conv_in)), (g,))
    if not need_dx:
        return None
    if g.dtype == torch.bfloat16 and l.conv_in % 64 == 0:
        return ops.convT4x4s2(g, l.wd, None, l.conv_in, defer=defer)     # data gradient by sub-pixel phase, plain output
    return ops.depth_to_space_unpad1_t(ops.convkxk(g, l.wd, None, 4 * l.conv_in, 2, 1))


def _conv_t2_fwd(l: _Layer, x: torch.Tensor, defer: bool = False):
    """nn.ConvTranspose2d(k=4, s=2, p=1): [B,h,w,Cin] -> [B,2h,2w,Cout]."""
    bias = None if l.bias is None else l.bias.detach()
    if x.dtype == torch.bfloat16 and l.conv_out % 64 == 0:
        return ops.convT4x4s2(x, l.wd, bias, l.conv_out, defer=defer)
    # space-to-depth form: the four sub-pixel channel groups of the output share the layer's bias (bias_mod)
    return ops.depth_to_space_unpad1_t(ops.convkxk(x, l.wd, bias, 4 * l.conv_out, 2, 1, bias_mod=l.conv_out))


def _conv_t2_bwd(l: _Layer, g: torch.Tensor, x: torch.Tensor, gw: torch.Tensor, need_dx: bool = True):
    # weight gradient with the roles exchanged (the result is nn.ConvTranspose2d's [Cin][Cout][4][4]); data gradient =
    # the stride-2 convolution of g with the forward operand
    if ops.fused_s2_ok(g.dtype, g.shape[3]):
        engine.run_on_side(lambda: ops.convkxk_wgrad(x, g, gw, 2, x_plain=True), (g,))
        return ops.conv4x4s2(g, l.wf, None, l.conv_in) if need_dx else None
    gs = ops.space_to_depth_pad1_t(g)
    engine.run_on_side(lambda: ops.convkxk_wgrad(x, gs, gw, 2), (gs,))
    return ops.convkxk(gs, l.wf, None, l.conv_in, 2, 0) if need_dx else None


def _conv_s1_bwd(l: _Layer, g: torch.Tensor, x: torch.Tensor, gw: torch.Tensor, want_w: bool = True) -> torch.Tensor:
    if want_w:
        engine.run_on_side(lambda: ops.convkxk_wgrad(g, x, gw, 4), (g,))
    return ops.convkxk(g, l.wd, None, l.conv_in, 4, 2)


@dataclass
class _GCtx:
    xs: List[torch.Tensor] = field(default_factory=list)        # what every down layer's weight gradient reads (its input)
    raw: List[Optional[torch.Tensor]] = field(default_factory=list)   # conv output ahead of a norm (None: no norm)
    act: List[torch.Tensor] = field(default_factory=list)       # down-layer activations a_i
    stats: List[Optional[torch.Tensor]] = field(default_factory=list)
    cat: List[torch.Tensor] = field(default_factory=list)       # input of up layer j ([relu(skip) | u_{j-1}])
    uraw: List[torch.Tensor] = field(default_factory=list)      # transposed-conv output ahead of its norm
    ustats: List[torch.Tensor] = field(default_factory=list)
    h: torch.Tensor = None                                      # pre-tanh output [B,H,W,8]


class Pix2PixTrainer:
    """Fused G + D training step for ``Pix2PixGenerator`` / ``PatchGANDiscriminator`` (their parameters become views of
    this trainer's flat buffers; ``state_dict`` / ``load_state_dict`` of the modules keep working).

    ``precision``: "bf16" (throughput) or "fp32" (three-way-split MFMA parity mode, checked at 1e-3 against the oracle).
    """

    def __init__(self, G: Pix2PixGenerator, D: PatchGANDiscriminator, lr: float = 2e-4,
                 betas: Tuple[float, float] = (0.5, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 lambda_l1: float = 100.0, precision: str = "bf16", bucket_mb: float = 4.0, process_group=None,
                 sync_loss: bool = True, max_bucket_mb: float = 16.0, sharded_optimizer: bool = False,
                 graph: bool = False):
        """``graph``: ``step()`` replays one captured hipGraph per G + D step (~200 launches on two streams); the first
        step of a batch shape runs eagerly, the second captures; bit-equal to the eager step (see CFMTrainer)."""
        dev = next(G.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("stain2stain_amd: Pix2PixTrainer needs the networks on a GPU (HIP-only implementation)")
        if precision not in ("bf16", "fp32"):
            raise ValueError("precision must be 'bf16' or 'fp32'")
        self.G, self.D = G, D
        _check_norms(G)
        _check_norms(D)
        self.dtype = torch.bfloat16 if precision == "bf16" else torch.float32
        self.lr, self.betas, self.eps, self.wd, self.lambda_l1 = lr, tuple(betas), eps, weight_decay, lambda_l1
        self.pg, self.sync_loss = process_group, sync_loss
        n = len(G.downs)
        self.n = n
        self.g_down = [_Layer(f"downs.{i}", "s2", m.weight, m.bias) for i, m in enumerate(G.downs)]
        self.g_up = [_Layer(f"ups.{j}", "t2", m.weight, m.bias) for j, m in enumerate(G.ups)]
        self.d_layers = [_Layer(name, kind, m.weight, m.bias) for name, kind, m in D.conv_layers()]

        def group(l: _Layer):
            return [(l.name + ".weight", l.weight)] + ([(l.name + ".bias", l.bias)] if l.bias is not None else [])

        # backward-completion order: last up layer first ... first down layer last; D: c5 ... c1
        self.pG = FlatParams([group(l) for l in reversed(self.g_up)] + [group(l) for l in reversed(self.g_down)],
                             bucket_mb, process_group, max_bucket_mb, sharded_optimizer)
        self.pD = FlatParams([group(l) for l in reversed(self.d_layers)], bucket_mb, process_group, max_bucket_mb,
                             sharded_optimizer)
        self.packG = _Packer(self.g_down + self.g_up, self.dtype)
        self.packD = _Packer(self.d_layers, self.dtype)
        self.in_channels, self.out_channels = G.in_channels, G.out_channels
        # weight gradients overlap the bandwidth-bound backward passes on a second HIP stream (engine.run_on_side);
        # S2S_WGRAD_STREAM=0 keeps everything on one stream, overlap_wgrad = False does so for a single step
        self._side = ops.side_stream_for(dev)
        self.overlap_wgrad = True
        # generator update inside its backward pass (S2S_P2P_OPT_IN_BWD=1): a layer's Adam slice + repack go onto the side
        # stream right behind its weight gradient instead of 0.4 ms of bandwidth-bound work (54 M parameters) at the end of
        # the step.  Element-wise identical to the end-of-step update; only without a gradient exchange.  Off by default:
        # measured 3.90 vs 3.86 ms/step -- the step is bound by total GPU work, not by the compute stream's length (the
        # overlapped Adam slows the convolutions it overlaps by as much as it saves), and it costs 30 more launches.
        self.opt_in_backward = os.environ.get("S2S_P2P_OPT_IN_BWD", "0") == "1"
        self.last: Dict[str, torch.Tensor] = {}
        self.graph = graph
        self._captured, self._warm_key = None, None
        self._hyper = None                     # ops.AdamHyperRing, created with the first captured step

    def optimizer_state_dict(self) -> Dict:
        """Both Adam states (by parameter name) + hyper-parameters; a collective with ``sharded_optimizer`` (FlatParams.state_dict)."""
        return {"G": self.pG.state_dict(), "D": self.pD.state_dict(),
                "hyper": {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.wd}}

    def load_optimizer_state_dict(self, sd: Dict) -> None:
        self.pG.load_state_dict(sd["G"])
        self.pD.load_state_dict(sd["D"])
        h = sd.get("hyper", {})
        self.lr, self.betas = h.get("lr", self.lr), tuple(h.get("betas", self.betas))
        self.eps, self.wd = h.get("eps", self.eps), h.get("weight_decay", self.wd)

    def _mark(self, fp: FlatParams, group: int) -> None:
        """Group ``group`` of ``fp`` has its gradients enqueued (bias gradients on the compute stream, the weight gradient
        on the side stream): exchange the buckets it closes, ordered behind both streams."""
        fp.bucketer.mark_ready_ordered(group, engine.side_stream)

    def _join(self) -> None:
        engine.join_side()
        engine.side_stream = None

    # ------------------------------------------------------------------------------------------------------------
    # generator
    # ------------------------------------------------------------------------------------------------------------
    def g_forward(self, src: torch.Tensor) -> _GCtx:
        """src: NCHW fp32 [B, in_channels, H, W] -> ctx with ctx.h = pre-tanh output [B,H,W,8]."""
        n, dt = self.n, self.dtype
        B, _, H, W = src.shape
        if H % (1 << n) or W % (1 << n):
            raise ValueError(f"tile size must be a multiple of {1 << n} for {n} down-sampling layers")
        dev = src.device
        ctx = _GCtx()
        ch = [l.conv_out for l in self.g_down]
        # concatenation buffers of the up path: cat[j] = [relu(a_{n-1-j}) | u_{j-1}], j >= 1
        ctx.cat = [None] * n
        for j in range(1, n):
            i = n - 1 - j
            ctx.cat[j] = torch.empty((B, H >> (i + 1), W >> (i + 1), 2 * ch[i]), dtype=dt, device=dev)
        x = ops.p2p_pack_input(src, None, torch.empty((B, H, W, 8), dtype=dt, device=dev))
        for i, l in enumerate(self.g_down):
            skip = ctx.cat[n - 1 - i][..., :ch[i]] if i < n - 1 else None
            small = i > 0 and ops.convsm_wins(dt, 1, B, x.shape[1] // 2, x.shape[2] // 2, x.shape[3], l.conv_out)
            if i == 0:                                    # LeakyReLU, no norm
                a, saved = _conv_s2_fwd(l, x, act=True, slope=LRELU, out2=skip)
                ctx.raw.append(None); ctx.stats.append(None)
            elif i == n - 1:                              # innermost: ReLU, no norm
                if small:                                 # (one sample-complete launch, dead taps skipped)
                    a, saved = ops.convsm_fwd(1, x, l.wf, _b(l), l.conv_out, norm=False, act=True, slope=0.0)[0], x
                else:
                    a, saved = _conv_s2_fwd(l, x, act=True, slope=0.0)
                ctx.raw.append(None); ctx.stats.append(None)
            elif small:                                   # inner level: convolution + norm + both outputs in one launch
                a, raw, st = ops.convsm_fwd(1, x, l.wf, _b(l), l.conv_out, norm=True, slope=LRELU, out2=skip)
                saved = x
                ctx.stats.append(st)
                ctx.raw.append(raw)
            else:
                raw, saved = _conv_s2_fwd(l, x, defer=FOLD_IN_NORM)
                if isinstance(raw, ops.SplitSum):        # inner level: the norm folds the split-K slabs itself
                    a = torch.empty(raw.shape, dtype=dt, device=dev)
                    st, raw = ops.instnorm_lrelu_fwd2_split(raw, LRELU, a, skip)
                else:
                    a = torch.empty_like(raw)
                    st = ops.instnorm_lrelu_fwd2(raw, LRELU, a, skip)
                ctx.stats.append(st)
                ctx.raw.append(raw)
            ctx.xs.append(saved)
            ctx.act.append(a)
            x = a
        for j, l in enumerate(self.g_up):
            xin = x if j == 0 else ctx.cat[j]
            if j < n - 1 and ops.convsm_wins(dt, 2, B, xin.shape[1], xin.shape[2], xin.shape[3], l.conv_out):
                C = l.conv_out                            # transposed convolution + norm + ReLU into the concat buffer
                _, hraw, st = ops.convsm_fwd(2, xin, l.wd, _b(l), C, norm=True, slope=0.0, out=ctx.cat[j + 1][..., C:])
                ctx.ustats.append(st)
                ctx.uraw.append(hraw)
                continue
            hraw = _conv_t2_fwd(l, xin, defer=FOLD_IN_NORM and j < n - 1)
            if j < n - 1:
                C = l.conv_out
                if isinstance(hraw, ops.SplitSum):
                    st, hraw = ops.instnorm_lrelu_fwd2_split(hraw, 0.0, ctx.cat[j + 1][..., C:])
                else:
                    st = ops.instnorm_lrelu_fwd2(hraw, 0.0, ctx.cat[j + 1][..., C:])
                ctx.ustats.append(st)
                ctx.uraw.append(hraw)
            else:
                ctx.h = hraw
        return ctx

    def _update_group(self, grp: int, l: _Layer, hyper_dev) -> None:
        """Adam + repack of generator layer ``l`` (parameter group ``grp``), behind its gradients on both streams."""
        side = engine.side_stream
        if side is None:
            self.pG.adam_group(grp, self.lr, self.betas, self.eps, self.wd, hyper_dev)
            self.packG.repack_layer(l.name)
            return
        side.wait_stream(torch.cuda.current_stream())     # bias gradient; the data gradient that reads the packed operand
        with torch.cuda.stream(side):
            self.pG.adam_group(grp, self.lr, self.betas, self.eps, self.wd, hyper_dev)
            self.packG.repack_layer(l.name)

    def g_backward(self, ctx: _GCtx, dh: torch.Tensor, update: bool = False, hyper_dev=None) -> None:
        """dh: gradient wrt the pre-tanh output.  Gradients go to the flat buffer; buckets are launched as they close.
        ``update``: every layer's Adam step + repack follow its gradients at once (see ``opt_in_backward``)."""
        n, gr, bk = self.n, self.pG.grads, self.pG.bucketer
        bk.start_step()
        if update and hyper_dev is None:
            self.pG.step_count += 1
        ch = [l.conv_out for l in self.g_down]
        grp = 0
        g = dh
        B, dt = dh.shape[0], dh.dtype
        upg = [None] * n           # gradient wrt up layer j's OUTPUT ahead of its norm backward (the second half of cat[j + 1])
        skipg = [None] * n         # gradient wrt the skip half of cat[j] (second gradient of down layer n - 1 - j's norm)
        g_is_dz = False            # g already went through the norm backward of the layer it now belongs to
        for j in range(n - 1, -1, -1):
            l = self.g_up[j]
            if j == n - 1:
                if l.bias is not None:
                    ops.channel_sum_into(g, gr[l.name + ".bias"])
            elif not g_is_dz:
                g = ops.instnorm_lrelu_bwd2(upg[j], None, ctx.uraw[j], ctx.ustats[j], 0.0)
            g_is_dz = False
            xin = ctx.act[n - 1] if j == 0 else ctx.cat[j]
            gw = gr[l.name + ".weight"]
            # data gradient = the stride-2 convolution of g with the forward operand; on the inner levels one launch that
            # ends in the norm backward of up layer j - 1 (whose output is the second half of this layer's input)
            hin, win = xin.shape[1], xin.shape[2]
            if ops.convsm_wins(dt, 1, B, hin, win, g.shape[3], l.conv_in):
                engine.run_on_side(lambda x=xin, gg=g, gw=gw: ops.convkxk_wgrad(x, gg, gw, 2, x_plain=True), (g,))
                if j == 0:
                    _, g = ops.convsm_bwd(1, g, l.wf, l.conv_in, z=None, stats=None, g2=None, slope=0.0, bwd_c0=l.conv_in)
                else:
                    Cs = l.conv_in // 2
                    g, skipg[j] = ops.convsm_bwd(1, g, l.wf, l.conv_in, z=ctx.uraw[j - 1], stats=ctx.ustats[j - 1], g2=None,
                                                 slope=0.0, bwd_c0=Cs)
                    g_is_dz = True
            else:
                dx = _conv_t2_bwd(l, g, xin, gw)
                if j == 0:
                    g = dx
                else:
                    Cs = l.conv_in // 2
                    skipg[j], upg[j - 1] = dx[..., :Cs], dx[..., Cs:]
            self._mark(self.pG, grp)
            if update:
                self._update_group(grp, l, hyper_dev)
            grp += 1
        g_is_dz = False
        for i in range(n - 1, -1, -1):
            l = self.g_down[i]
            skip_g = skipg[n - 1 - i] if i < n - 1 else None
            if g_is_dz:
                pass
            elif i == n - 1:
                g = ops.p2p_act_bwd(g, None, ctx.act[i], 0.0, gr.get(l.name + ".bias"))
            elif i == 0:
                g = ops.p2p_act_bwd(g, skip_g, ctx.act[0], LRELU, gr.get(l.name + ".bias"))
            else:
                g = ops.instnorm_lrelu_bwd2(g, skip_g, ctx.raw[i], ctx.stats[i], LRELU)
            g_is_dz = False
            gw = gr[l.name + ".weight"]
            # (layer i - 1 has a norm for i >= 2: the inner levels' data gradient ends in it)
            if i >= 2 and ops.convsm_wins(dt, 2, B, g.shape[1], g.shape[2], g.shape[3], l.conv_in):
                engine.run_on_side(lambda gg=g, x=ctx.xs[i], gw=gw: ops.convkxk_wgrad(gg, x, gw, 2, x_plain=True), (g,))
                g, _ = ops.convsm_bwd(2, g, l.wd, l.conv_in, z=ctx.raw[i - 1], stats=ctx.stats[i - 1],
                                      g2=skipg[n - i], slope=LRELU)
                g_is_dz = True
            else:
                g = _conv_s2_bwd(l, g, ctx.xs[i], gw, need_dx=(i > 0), defer=FOLD_IN_NORM and i > 1)
            self._mark(self.pG, grp)
            if update:
                self._update_group(grp, l, hyper_dev)
            grp += 1

    # ------------------------------------------------------------------------------------------------------------
    # discriminator
    # ------------------------------------------------------------------------------------------------------------
    def d_forward(self, d_in: torch.Tensor):
        """d_in [N,H,W,8] = [src | target-or-fake | 0 0] -> (logits [N,h,w,8] (channel 0), saved): ``saved[k]`` =
        (what layer k's weight gradient reads, conv output ahead of the norm or None, norm statistics or None,
        activation or None) for the layers c1 ... cK of the PatchGAN (first: LeakyReLU without a norm; last: logits)."""
        L, saved, x = self.d_layers, [], d_in
        for k, l in enumerate(L):
            if k == len(L) - 1:
                z = ops.convkxk(x, l.wf, l.bias.detach(), l.conv_out, 4, 1)
                saved.append((x, None, None, None))
                return z, saved
            if k == 0:
                a, xs = _conv_s2_fwd(l, x, act=True, slope=LRELU)
                saved.append((xs, None, None, a))
            else:
                if l.kind == "s2":
                    r, xs = _conv_s2_fwd(l, x)
                else:
                    r, xs = ops.convkxk(x, l.wf, l.bias.detach(), l.conv_out, 4, 1), x
                a = torch.empty_like(r)
                st = ops.instnorm_lrelu_fwd2(r, LRELU, a)
                saved.append((xs, r, st, a))
            x = a

    def d_backward(self, saved, dz: torch.Tensor, want_w: bool, need_input_grad: bool):
        """want_w: form the discriminator's weight gradients (D update) -- the generator update passes False and only
        pulls the data gradient through the frozen discriminator.  Returns d(d_in) or None."""
        L = self.d_layers
        gr, bk = self.pD.grads, self.pD.bucketer
        if want_w:
            bk.start_step()
        g = dz
        for grp, k in enumerate(range(len(L) - 1, -1, -1)):
            l = L[k]
            xs, r, st, a = saved[k]
            if k == len(L) - 1:
                if want_w:
                    ops.channel_sum_into(g, gr[l.name + ".bias"])
            elif k == 0:
                g = ops.p2p_act_bwd(g, None, a, LRELU, gr[l.name + ".bias"] if want_w else None)
            else:
                g = ops.instnorm_lrelu_bwd2(g, None, r, st, LRELU)
            if l.kind == "s1":
                g = _conv_s1_bwd(l, g, xs, gr[l.name + ".weight"], want_w)
            else:
                g = _conv_s2_bwd(l, g, xs, gr[l.name + ".weight"], k > 0 or need_input_grad, want_w)
            if want_w:
                self._mark(self.pD, grp)
        return g

    # ------------------------------------------------------------------------------------------------------------
    # the step
    # ------------------------------------------------------------------------------------------------------------
    def losses_and_grads(self, src: torch.Tensor, tgt: torch.Tensor, update: bool = True, want_fake: bool = False,
                         hyper_dev: Optional[torch.Tensor] = None):
        """One G + D evaluation.  ``update=True`` is the training step (D is updated before the generator's pass through
        it, as in pix2pix_step); ``update=False`` leaves both networks untouched and evaluates the two losses and both
        gradient sets at the CURRENT parameters (what ``(loss_D + loss_G).backward()`` of pix2pix_losses yields), for the
        parity tests.  Returns (losses, fake): ``losses`` = device float[8] {BCE(D(real), 1), BCE(D(fake), 0),
        BCE(D(fake), 1), 0, L1(fake, tgt), 0, 0, 0} (no host synchronisation; ``loss_values`` turns it into loss_D / loss_G),
        ``fake`` = NCHW fp32 or None."""
        B, C, H, W = tgt.shape
        dev, dt = src.device, self.dtype
        src, tgt = src.contiguous().float(), tgt.contiguous().float()
        losses = torch.zeros((8,), dtype=torch.float32, device=dev)
        ops.PROFILE_TAG = "G"
        gctx = self.g_forward(src)
        ops.PROFILE_TAG = "D"
        d_in = torch.empty((2 * B, H, W, 8), dtype=dt, device=dev)          # [real pairs | fake pairs]
        ops.p2p_pack_input(src, tgt, d_in[:B])
        fake = torch.empty((B, C, H, W), dtype=torch.float32, device=dev) if want_fake else None
        ops.p2p_tanh_l1_fwd(gctx.h, src, tgt, d_in[B:], fake, l1_out=losses[4:5])
        # ---- discriminator update: real and detached fake as one batch of 2B samples ----
        z, saved = self.d_forward(d_in)
        npatch = z.shape[1] * z.shape[2]
        _, dz = ops.p2p_bce_logits(z, B, 0.5 / (B * npatch), 0.5 / (B * npatch), out=losses[0:2])
        engine.side_stream = self._side if self.overlap_wgrad else None
        self.d_backward(saved, dz, want_w=True, need_input_grad=False)
        self._join()
        if update:
            self.pD.adam(self.lr, self.betas, self.eps, self.wd, hyper_dev)
            self.packD.repack()
            zg, saved_g = self.d_forward(d_in[B:])                          # through the UPDATED discriminator
        else:
            self.pD.bucketer.wait_all()
            zg, saved_g = z[B:], [tuple(None if t is None else self._second_half(t, B) for t in lay) for lay in saved]
        # ---- generator update ----
        _, dzg = ops.p2p_bce_logits(zg, B, 1.0 / (B * npatch), 0.0, out=losses[2:4])
        gd = self.d_backward(saved_g, dzg, want_w=False, need_input_grad=True)
        dh = ops.p2p_tanh_l1_bwd(gctx.h, tgt, gd, self.lambda_l1 / (B * C * H * W))
        ops.PROFILE_TAG = "G"
        engine.side_stream = self._side if self.overlap_wgrad else None
        in_bwd = update and self.opt_in_backward and not self.pG.bucketer.enabled
        self.g_backward(gctx, dh, update=in_bwd, hyper_dev=hyper_dev)
        self._join()
        if update and not in_bwd:
            self.pG.adam(self.lr, self.betas, self.eps, self.wd, hyper_dev)
            self.packG.repack()
        elif not update:
            self.pG.bucketer.wait_all()
        ops.PROFILE_TAG = ""
        return losses, fake

    @staticmethod
    def _second_half(t: torch.Tensor, B: int) -> torch.Tensor:
        # saved tensors are per-sample along dim 0, except the InstanceNorm statistics [4][N][C]
        return t[:, B:].contiguous() if (t.dim() == 3 and t.shape[0] == 4) else t[B:]

    def loss_values(self, losses: torch.Tensor) -> Tuple[float, float]:
        """(loss_D, loss_G) = (0.5 (BCE_real + BCE_fake), BCE_gan + lambda L1) from a ``losses`` vector (one host read)."""
        v = losses.detach().cpu().tolist()
        return 0.5 * (v[0] + v[1]), v[2] + self.lambda_l1 * v[4]

    def step(self, src: torch.Tensor, tgt: torch.Tensor) -> torch.Tensor:
        """One training step on this rank's shard of the global batch (the reference shards a batch as
        ``batch_size // world_size`` per rank, src/data/paired_data_module.py:273-278); returns the ``losses`` vector
        (rank means when ``sync_loss``), still on the device."""
        with ops.workspace_owner(self):          # split-K / weight-gradient slabs belong to this trainer (ops._workspace)
            if self.graph and ops._PROFILE is None:
                return self._step_graphed(src, tgt)
            return self._step_body(src, tgt, None)

    def _step_body(self, src, tgt, hyper_dev) -> torch.Tensor:
        losses, _ = self.losses_and_grads(src, tgt, update=True, hyper_dev=hyper_dev)
        if self.sync_loss:
            work = all_reduce_mean_scalar(losses, self.pg)
            if work is not None:
                work.wait()
                ops.axpy_(losses, losses, 1.0 / dist.get_world_size(self.pg) - 1.0)     # x += (1/w - 1) x
        return losses

    def _step_graphed(self, src: torch.Tensor, tgt: torch.Tensor) -> torch.Tensor:
        key = (tuple(src.shape), tuple(tgt.shape), src.device, self.overlap_wgrad)
        if self.pG.bucketer.enabled and dist.get_backend(self.pg) != "nccl":
            raise RuntimeError("stain2stain_amd: graph=True captures the gradient exchange; only RCCL ('nccl') "
                               "collectives can be captured")
        if self._hyper is None:
            self._hyper = ops.AdamHyperRing(src.device)
        if self._captured is None or self._captured[0] != key:
            if self._warm_key != key:       # first step of this shape: eager (module load, LDS attributes, workspaces)
                self._warm_key, self._captured = key, None
                return self._step_body(src, tgt, None)
            s_src = torch.empty_like(src, dtype=torch.float32).contiguous()
            s_tgt = torch.empty_like(tgt, dtype=torch.float32).contiguous()
            graph = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with ops.capture_graph(graph):
                losses = self._step_body(s_src, s_tgt, self._hyper.dev)
            self._captured = (key, graph, s_src, s_tgt, losses)
        _, graph, s_src, s_tgt, losses = self._captured
        s_src.copy_(src); s_tgt.copy_(tgt)
        self.pD.step_count += 1
        self.pG.step_count += 1
        if self.pD.step_count != self.pG.step_count:
            raise RuntimeError("graph=True: the two optimisers must have taken the same number of steps")
        self._hyper.push(ops.adam_hyper(self.pG.step_count, self.lr, self.betas[0], self.betas[1], self.eps, self.wd,
                                        self.pG.bucketer.grad_scale))
        graph.replay()
        return losses.clone()

    def close(self) -> None:
        """Drop the captured graph, its static buffers and the pinned Adam-scalar ring now (see CFMTrainer.close)."""
        self._captured, self._warm_key, self._hyper = None, None, None
        ops.release_workspaces(self)

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    @torch.no_grad()
    def generate(self, src: torch.Tensor) -> torch.Tensor:
        """fake = G(src) (training-mode statistics are the only ones InstanceNorm has), NCHW fp32."""
        B, _, H, W = src.shape
        src = src.contiguous().float()
        ctx = self.g_forward(src)
        d_in = torch.empty((B, H, W, 8), dtype=self.dtype, device=src.device)
        fake = torch.empty((B, self.out_channels, H, W), dtype=torch.float32, device=src.device)
        zeros = torch.zeros_like(fake)
        ops.p2p_tanh_l1_fwd(ctx.h, src[:, :self.out_channels].contiguous(), zeros, d_in, fake)
        return fake


# ----------------------------------------------------------------------------------------------------------------------
# the same passes without a trainer: what the autograd face of the two modules runs (pix2pix.py)
# ----------------------------------------------------------------------------------------------------------------------
class _NoExchange:
    enabled = False

    def start_step(self) -> None:
        pass

    def mark_ready_ordered(self, group_index, side) -> None:
        pass


class _GradSet:
    def __init__(self):
        self.grads: Dict[str, torch.Tensor] = {}
        self.bucketer = _NoExchange()


class NetRunner:
    """``g_forward`` / ``g_backward`` / ``d_forward`` / ``d_backward`` of ``Pix2PixTrainer`` on ONE module's own parameters
    (no flat buffers, no optimiser, no side stream): ``Pix2PixGenerator.forward`` and ``PatchGANDiscriminator.forward`` are
    each a single autograd node over these passes, so the module face launches the same HIP kernels as the fused trainer
    -- LeakyReLU / ReLU in the conv epilogues, ReLU'd skips written into the concatenation buffers by the norm pass, no
    torch activation, ``cat`` or ``pad``.  The packed MFMA operands are refreshed whenever a weight's version changes
    (an optimiser step, ``load_state_dict``)."""
    g_forward = Pix2PixTrainer.g_forward
    g_backward = Pix2PixTrainer.g_backward
    d_forward = Pix2PixTrainer.d_forward
    d_backward = Pix2PixTrainer.d_backward
    _mark = Pix2PixTrainer._mark

    def __init__(self, net, precision: str = "bf16"):
        if precision not in ("bf16", "fp32"):
            raise ValueError("precision must be 'bf16' or 'fp32'")
        self.net = net
        self.dtype = torch.bfloat16 if precision == "bf16" else torch.float32
        self.is_generator = isinstance(net, Pix2PixGenerator)
        self.pG, self.pD = _GradSet(), _GradSet()
        self._key, self._packer = None, None
        self.overlap_wgrad = False

    def _layers(self) -> List[_Layer]:
        return self.g_down + self.g_up if self.is_generator else self.d_layers

    def refresh(self) -> None:
        """(Re)build the layer list when the parameters moved, repack when a weight changed."""
        net = self.net
        ptrs = tuple(p.data_ptr() for p in net.parameters())
        vers = tuple(p._version for p in net.parameters())
        if self._key is None or self._key[0] != ptrs:
            if next(net.parameters()).device.type != "cuda":
                raise RuntimeError("stain2stain_amd: the pix2pix networks run on the GPU only (no CPU fallback)")
            _check_norms(net)
            if self.is_generator:
                self.n = len(net.downs)
                self.g_down = [_Layer(f"downs.{i}", "s2", m.weight, m.bias) for i, m in enumerate(net.downs)]
                self.g_up = [_Layer(f"ups.{j}", "t2", m.weight, m.bias) for j, m in enumerate(net.ups)]
            else:
                self.d_layers = [_Layer(name, kind, m.weight, m.bias) for name, kind, m in net.conv_layers()]
            self._packer = _Packer(self._layers(), self.dtype)                # packs
        elif self._key[1] != vers:
            self._packer.repack()
        self._key = (ptrs, vers)

    def new_grads(self) -> Dict[str, torch.Tensor]:
        """Fresh gradient tensors by parameter name (autograd keeps what ``backward`` returns): weights are written whole
        by their kernels; biases start at zero -- the one in front of an InstanceNorm has an exactly zero gradient and no
        kernel."""
        out = {}
        for l in self._layers():
            out[l.name + ".weight"] = torch.empty_like(l.weight, memory_format=torch.contiguous_format)
            if l.bias is not None:
                out[l.name + ".bias"] = torch.zeros_like(l.bias)
        return out
