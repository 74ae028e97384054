"""Forward / backward schedules of the U-Net over the HIP ops (no autograd, no torch math).

The functions here walk the parameter containers defined in ``components.py`` and issue the kernel
sequence for

  * ``encoder_forward`` / ``encoder_backward``   SharedEncoder.forward (reference
    src/models/components/shared_encoder.py:75-104) and its autograd backward
  * ``decoder_forward`` / ``decoder_backward``   FlowMatchingDecoder.forward (task_decoders.py:102-134)

Every intermediate is an NHWC tensor in the compute dtype (bf16, or fp32 for the split-MFMA parity
mode) allocated from torch's caching allocator; a forward returns a context object that owns what
its backward needs, so two forwards may be alive at once (the multitask module runs the encoder
twice per step, conditional_flow_matching_multitask.py:228,236).

Gradients are written into ``grads[name]`` (fp32, parameter-shaped, e.g. views of one flat bucket
buffer); with ``accumulate=True`` they are added to what is there.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import contextlib

import torch
import torch.distributed as dist

from . import ops

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# ------------------------------------------------------------------------------------------------
# SyncBatchNorm
# ------------------------------------------------------------------------------------------------
class SyncBNExchange:
    """The two exchanges of ``torch.nn.SyncBatchNorm`` (what Lightning's ``sync_batchnorm: True`` of the reference's
    ``trainer=ddp`` preset, configs/trainer/ddp.yaml:9, wraps every BatchNorm2d in): the per-channel (sum, sum of squares)
    of the forward and (sum dy, sum dy * xhat) / N of the backward, summed over the ranks of ``process_group``.  Each is
    one small blocking all-reduce (2 C floats) per layer, in stream order on RCCL; with a ``gloo`` group the 2 C floats
    are staged through the host.  Every rank must hold the same number of elements per channel."""

    def __init__(self, process_group=None):
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("stain2stain_amd: SyncBatchNorm needs an initialised torch.distributed process group")
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self._host = dist.get_backend(process_group) != "nccl"

    def __call__(self, t: torch.Tensor) -> None:
        if self._host and t.is_cuda:
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.pg)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)


sync_bn: Optional[SyncBNExchange] = None      # set for the duration of a training step (sync_batchnorm below)
_module_exchanges: Dict[int, SyncBNExchange] = {}


def _module_exchange(bn: torch.nn.Module) -> Optional[SyncBNExchange]:
    """A BatchNorm container that has been converted by ``torch.nn.SyncBatchNorm.convert_sync_batchnorm`` (Lightning's
    ``sync_batchnorm: True``, configs/trainer/ddp.yaml:9) synchronises over its own process group, as the torch module
    would: in training, when torch.distributed is up and the group has more than one rank (S2S_FORCE_DDP=1: any)."""
    if not isinstance(bn, torch.nn.SyncBatchNorm) or not (dist.is_available() and dist.is_initialized()):
        return None
    pg = bn.process_group
    import os
    if dist.get_world_size(pg) < 2 and os.environ.get("S2S_FORCE_DDP") != "1":
        return None
    key = id(pg) if pg is not None else 0
    ex = _module_exchanges.get(key)
    if ex is None or ex.pg is not pg:          # (an id can be re-used by a later group: the entry holds its own group)
        ex = _module_exchanges[key] = SyncBNExchange(pg)
    return ex


@contextlib.contextmanager
def sync_batchnorm(exchange: Optional[SyncBNExchange]):
    """Train-mode BatchNorm layers whose FORWARD runs inside this context take their statistics over all ranks of
    ``exchange``; the layer remembers the exchange, so its backward (autograd's, whenever it runs) uses the same one."""
    global sync_bn
    prev, sync_bn = sync_bn, exchange
    try:
        yield
    finally:
        sync_bn = prev


# ------------------------------------------------------------------------------------------------
# parameter bundles
# ------------------------------------------------------------------------------------------------
mutation_epoch = [0]     # see ConvBN.eval_affine


class ConvBN:
    """One Conv3x3 + BatchNorm2d pair of a DoubleConv, with its packed MFMA weight cache."""

    def __init__(self, prefix: str, i_conv: str, i_bn: str, seq: torch.nn.Sequential):
        self.prefix = prefix            # e.g. "inc.double_conv"; parameter names are f"{prefix}.{i_conv}.weight" ...
        self.idx = (i_conv, i_bn)       # ("0", "1") or ("3", "4"): positions inside the reference's nn.Sequential
        self._seq = seq                 # looked up on every use: torch.nn.SyncBatchNorm.convert_sync_batchnorm (what
                                        # Lightning's sync_batchnorm: True calls) REPLACES the BatchNorm2d in the container
        self._pack: Dict[torch.dtype, Tuple] = {}
        self._pack_key: Dict[torch.dtype, Tuple] = {}
        self._eval_key, self._eval_ss = None, None

    @property
    def conv(self) -> torch.nn.Conv2d:
        return self._seq._modules[self.idx[0]]

    @property
    def bn(self) -> torch.nn.BatchNorm2d:
        return self._seq._modules[self.idx[1]]

    @property
    def cout(self) -> int:
        return self.conv.weight.shape[0]

    @property
    def cin(self) -> int:
        return self.conv.weight.shape[1]

    def packed(self, dtype: torch.dtype):
        """(forward, dgrad) MFMA layouts of the fp32 OIHW master weight; repacked when it changes."""
        w = self.conv.weight
        key = (w.data_ptr(), w._version, w.device)
        if self._pack_key.get(dtype) != key:
            old = self._pack.get(dtype)
            reuse = old if (old is not None and old[0].device == w.device) else None
            self._pack[dtype] = ops.pack_conv3x3(w.detach(), dtype, want_dgrad=True, out=reuse)
            self._pack_key[dtype] = key
        return self._pack[dtype]

    def invalidate(self) -> None:
        self._pack_key.clear()

    def eval_affine(self):
        """Folded eval-mode BatchNorm (scale, shift), cached until a parameter / running statistic changes.
        Kernels that write those tensors through raw pointers (train-mode finalize, the fused Adam) do not bump
        torch's version counters, so they bump ``mutation_epoch`` instead."""
        bn = self.bn
        ts = (bn.weight, bn.bias, bn.running_mean, bn.running_var)
        key = (mutation_epoch[0],) + tuple((t.data_ptr(), t._version) for t in ts)
        if self._eval_key != key:
            self._eval_ss = ops.bn_eval_prepare(bn.weight.detach(), bn.bias.detach(), bn.running_mean,
                                                bn.running_var, bn.eps)
            self._eval_key = key
        return self._eval_ss

    def ensure_buffers(self, dtype: torch.dtype):
        """Allocate (once) the packed buffers without filling them; returns (wf, wd)."""
        old = self._pack.get(dtype)
        w = self.conv.weight
        if old is None or old[0].device != w.device:
            L = ops._L()
            wf = torch.empty((L.s2s_pack_conv3x3_fwd_elems(self.cout, self.cin),), dtype=dtype, device=w.device)
            wd = torch.empty((L.s2s_pack_conv3x3_dgrad_elems(self.cout, self.cin),), dtype=dtype, device=w.device)
            self._pack[dtype] = (wf, wd)
            self._pack_key.pop(dtype, None)
        return self._pack[dtype]

    def mark_packed(self, dtype: torch.dtype) -> None:
        """The packed copies were just refreshed externally (batched pack) from the current master weight."""
        w = self.conv.weight
        self._pack_key[dtype] = (w.data_ptr(), w._version, w.device)


_REPACK_DESC: Dict[tuple, Tuple[torch.Tensor, int]] = {}


def repack_stale(cbs: Sequence[ConvBN], dtype: torch.dtype) -> None:
    """Refresh the packed MFMA operands of every layer of ``cbs`` whose master weight changed since it was packed, in ONE
    launch (ops.pack_conv3x3_batched) -- the module path after an ``optimizer.step()``: left to ``ConvBN.packed`` each of
    the 17 layers re-packs itself with a launch of its own at first use (0.21 ms per step against 0.05)."""
    stale = []
    for cb in cbs:
        w = cb.conv.weight
        if not w.is_cuda:              # (a CPU module: the first op raises "no CPU path")
            return
        if cb._pack_key.get(dtype) != (w.data_ptr(), w._version, w.device):
            stale.append(cb)
    if len(stale) < 2:
        return
    bufs = [cb.ensure_buffers(dtype) for cb in stale]
    # (every field of a descriptor row is part of the key: the allocator re-uses addresses across modules)
    key = (dtype,) + tuple((cb.conv.weight.data_ptr(), b[0].data_ptr(), b[1].data_ptr(), cb.cout, cb.cin)
                           for cb, b in zip(stale, bufs))
    hit = _REPACK_DESC.get(key)
    if hit is None:
        rows, start = [], 0
        for cb, (wf, wd) in zip(stale, bufs):
            rows.append([cb.conv.weight.data_ptr(), wf.data_ptr(), wd.data_ptr(), cb.cout, cb.cin, start])
            start += ((cb.cout + 31) // 32) * ((cb.cin + 31) // 32)
        if len(_REPACK_DESC) > 64:
            _REPACK_DESC.clear()
        hit = _REPACK_DESC[key] = (torch.tensor(rows, dtype=torch.int64, device=stale[0].conv.weight.device), start)
    ops.pack_conv3x3_batched(hit[0], hit[1], dtype)
    for cb in stale:
        cb.mark_packed(dtype)


@dataclass
class LayerCtx:
    """What one conv+BN+ReLU layer keeps for its backward."""
    x0: Optional[torch.Tensor]          # conv input (NHWC), or the NCHW image for the stem
    x1: Optional[torch.Tensor]
    raw: torch.Tensor                   # conv output
    act: torch.Tensor                   # ReLU output
    stats: torch.Tensor                 # [4, C]: mean, invstd, scale, shift
    count: Optional[int] = None         # elements per channel the statistics were taken over (None = the local B*H*W)
    sync: Optional["SyncBNExchange"] = None     # SyncBatchNorm: the exchange the forward used (its backward uses it too)


@dataclass
class EncCtx:
    dtype: torch.dtype
    x_nchw: torch.Tensor
    layers: List[Tuple[LayerCtx, LayerCtx]] = field(default_factory=list)   # per level: (conv1, conv2)
    feats: List[torch.Tensor] = field(default_factory=list)                 # act of conv2 per level


@dataclass
class DecCtx:
    dtype: torch.dtype
    t_emb: torch.Tensor
    h1: torch.Tensor = None             # time_mlp.0 pre-activation
    a1: torch.Tensor = None             # SiLU output
    h2: torch.Tensor = None             # time_mlp.2 output
    lows: List[torch.Tensor] = field(default_factory=list)    # input of each up-sampling
    ups: List[torch.Tensor] = field(default_factory=list)
    skips: List[torch.Tensor] = field(default_factory=list)
    layers: List[Tuple[LayerCtx, LayerCtx]] = field(default_factory=list)
    v: torch.Tensor = None


def _stem_as_conv(x_nchw: torch.Tensor, dtype: torch.dtype) -> bool:
    """bf16 mode, <= 8 image channels: the stem runs as a generic conv3x3 on an 8-channel NHWC bf16 copy of the image
    (one 16 MB pack launch), i.e. at 256^2 on conv3x3_stage_kernel with its 3 -> 64 filter resident in LDS.
    stem_mfma_kernel builds its im2col patch per 16 x 16 tile with ~1850 vector instructions per wave (rocprofv3
    SQ_INSTS_VALU, profiles/r04_b): 109 us for a layer whose output is 134 MB.  fp32 parity mode keeps that kernel."""
    return dtype == torch.bfloat16 and x_nchw.shape[1] <= 8


def _stem_image(x_nchw: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    B, _, H, W = x_nchw.shape
    return ops.p2p_pack_input(x_nchw, None, torch.empty((B, H, W, 8), dtype=dtype, device=x_nchw.device))


def _conv_bn_relu(cb: ConvBN, x0, x1, dtype, training: bool, want_pool: bool = False, stem: bool = False):
    """Returns (LayerCtx or None, act, pooled)."""
    bn = cb.bn
    bias = cb.conv.bias.detach() if cb.conv.bias is not None else None
    if training:
        # A conv bias in front of a train-mode BatchNorm cancels in the normalised output (and its gradient is exactly
        # zero): the bf16 MFMA kernel is launched WITHOUT it -- its persistent form has no bias path -- and the bias only
        # enters the running mean (ops.bn_finalize, conv_bias).  The saved conv output and the saved mean are both those
        # of the unbiased convolution, so the backward is unchanged.  (fp32 parity mode and the stem add it as before.)
        late_bias = None
        if stem and _stem_as_conv(x0, dtype):
            raw, stat = ops.conv3x3(_stem_image(x0, dtype), None, cb.packed(dtype)[0], None, cb.cout, want_stats=True)
            late_bias = bias
        elif stem:
            raw, stat = ops.stem_fwd(x0, cb.conv.weight.detach(), bias, dtype, want_stats=True)
        elif dtype == torch.bfloat16 and bias is not None:
            raw, stat = ops.conv3x3(x0, x1, cb.packed(dtype)[0], None, cb.cout, want_stats=True)
            late_bias = bias
        else:
            raw, stat = ops.conv3x3(x0, x1, cb.packed(dtype)[0], bias, cb.cout, want_stats=True)
        count = raw.shape[0] * raw.shape[1] * raw.shape[2]
        sx = sync_bn if sync_bn is not None else _module_exchange(bn)
        if sx is not None:      # statistics over the global batch: all-reduce the per-channel sums, finalize on those
            sums = ops.bn_partial_sums(stat)
            sx(sums)
            stat = sums.view(2, -1, 1)
            count *= sx.world
        if count <= 1:      # torch.nn.functional.batch_norm raises the same way in training mode
            raise ValueError(f"Expected more than 1 value per channel when training, got input size "
                             f"{[raw.shape[0], raw.shape[3], raw.shape[1], raw.shape[2]]}")
        track = bn.track_running_stats and bn.running_mean is not None
        mutation_epoch[0] += 1
        st = ops.bn_finalize(stat, count, bn.weight.detach(), bn.bias.detach(),
                             bn.running_mean if track else None, bn.running_var if track else None,
                             bn.num_batches_tracked if track else None,
                             BN_MOMENTUM if bn.momentum is None else bn.momentum, bn.eps, conv_bias=late_bias)
        act, pooled = ops.bn_relu_apply(raw, st[2], st[3], want_pool=want_pool)
        return LayerCtx(x0, x1, raw, act, st, count if sx is not None else None, sx), act, pooled
    ss = cb.eval_affine()
    if stem and _stem_as_conv(x0, dtype):
        act, _ = ops.conv3x3(_stem_image(x0, dtype), None, cb.packed(dtype)[0], bias, cb.cout, scale=ss[0], shift=ss[1], relu=True)
        pooled = ops.maxpool2(act) if want_pool else None
    elif stem:
        raw, _ = ops.stem_fwd(x0, cb.conv.weight.detach(), bias, dtype, want_stats=False)
        act, pooled = ops.bn_relu_apply(raw, ss[0], ss[1], want_pool=want_pool)
    else:
        # eval-mode BatchNorm + ReLU folded into the MFMA epilogue: the conv writes the activation directly
        act, _ = ops.conv3x3(x0, x1, cb.packed(dtype)[0], bias, cb.cout, scale=ss[0], shift=ss[1], relu=True)
        pooled = ops.maxpool2(act) if want_pool else None
    return None, act, pooled


# ------------------------------------------------------------------------------------------------
# weight gradients on a side HIP stream
# ------------------------------------------------------------------------------------------------
# In the backward chain  BN-backward(L) -> { wgrad(L), dgrad(L) } -> BN-backward(L-1) -> ...  the weight gradient is
# a leaf: nothing downstream of it but the gradient exchange and Adam.  It is MFMA-bound, while the BatchNorm /
# up-sampling / pooling backward passes that follow on the critical path are HBM-bound, so the fused trainer issues the
# weight-gradient launches on a second stream (fork on an event after the layer's BN backward, join before the gradient
# bucket is exchanged / before Adam) and the two kinds of work share the chip.  ``side_stream`` is None on the
# autograd-module path (everything on the current stream).
side_stream: Optional[torch.cuda.Stream] = None


def run_on_side(fn, keep=()) -> None:
    """fn() on the side stream, ordered after everything enqueued on the current stream so far.  ``keep``: tensors the
    side-stream work reads that nothing else holds on to; they stay referenced until ``join_side`` has made the compute
    stream wait for the side stream, so the caching allocator cannot hand their memory out before the work has run.
    (``Tensor.record_stream`` would do too, but its blocks only return to the pool once the allocator has polled the
    side stream's events: with the host a step ahead of the GPU the pool grew by a hipMalloc -- a device
    synchronisation -- every other step, 11 -> 21 GB over 100 steps.)"""
    side = side_stream
    if side is None:
        fn()
        return
    ev = torch.cuda.Event()
    ev.record()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        fn()
    _side_keep.extend(t for t in keep if t is not None)


_side_keep: List[torch.Tensor] = []


def join_side(side: Optional[torch.cuda.Stream] = None) -> None:
    """The current stream waits for the side stream's work so far (before Adam reads the gradients); what that work
    read may be freed from here on."""
    s = side if side is not None else side_stream
    if s is not None:
        torch.cuda.current_stream().wait_stream(s)
    _side_keep.clear()


def _g(grads: Dict[str, torch.Tensor], name: str) -> torch.Tensor:
    try:
        return grads[name]
    except KeyError as e:
        raise RuntimeError(f"stain2stain_amd: no gradient buffer for parameter {name!r}") from e


def _conv_bn_relu_bwd(cb: ConvBN, lc: LayerCtx, g1, gp, grads, accumulate: bool, need_dx: bool, stem: bool = False):
    """Backward of conv -> BN -> ReLU.  g1/gp: gradient wrt the activation / wrt its 2x2 max-pool."""
    p = cb.prefix
    i_conv, i_bn = cb.idx
    dgamma = _g(grads, f"{p}.{i_bn}.weight")
    dbeta = _g(grads, f"{p}.{i_bn}.bias")
    dbias = _g(grads, f"{p}.{i_conv}.bias") if cb.conv.bias is not None else None
    draw = ops.bn_relu_bwd(g1, gp, lc.raw, lc.stats, cb.bn.weight.detach(), dgamma, dbeta, dbias, accumulate,
                           exchange=lc.sync, count_total=lc.count)
    dw = _g(grads, f"{p}.{i_conv}.weight")
    if stem:
        run_on_side(lambda: ops.stem_wgrad(draw, lc.x0, dw, None, accumulate), (draw,))
        return None
    run_on_side(lambda: ops.conv3x3_wgrad(draw, lc.x0, lc.x1, dw, accumulate), (draw,))
    if not need_dx:
        return None
    dx, _ = ops.conv3x3(draw, None, cb.packed(draw.dtype)[1], None, cb.cin)
    return dx


# ------------------------------------------------------------------------------------------------
# encoder
# ------------------------------------------------------------------------------------------------
def encoder_forward(blocks: Sequence[Tuple[ConvBN, ConvBN]], x_nchw: torch.Tensor, dtype: torch.dtype,
                    training: bool) -> EncCtx:
    """blocks[l] = (conv1, conv2) of level l (l = 0 is ``inc``).  Keeps every level's activation."""
    ctx = EncCtx(dtype, x_nchw)
    nlev = len(blocks)
    repack_stale([cb for l, pair in enumerate(blocks) for j, cb in enumerate(pair)
                  if (l, j) != (0, 0) or _stem_as_conv(x_nchw, dtype)], dtype)   # (the fp32-mode stem reads the master)
    inp = x_nchw
    for l, (c1, c2) in enumerate(blocks):
        lc1, a1, _ = _conv_bn_relu(c1, inp, None, dtype, training, stem=(l == 0))
        lc2, a2, pooled = _conv_bn_relu(c2, a1, None, dtype, training, want_pool=(l + 1 < nlev))
        ctx.layers.append((lc1, lc2))
        ctx.feats.append(a2)
        inp = pooled
    return ctx


def encoder_backward(blocks: Sequence[Tuple[ConvBN, ConvBN]], ctx: EncCtx, dfeats: Sequence[Optional[torch.Tensor]],
                     grads: Dict[str, torch.Tensor], accumulate: bool = False, on_group_done=None) -> None:
    """dfeats[l]: gradient wrt level l's output activation (NHWC, or None); the max-pool path between the
    levels is handled here.  ``on_group_done()`` fires after every conv + BatchNorm LAYER (deepest level first, second
    conv of a level before its first) once its parameter gradients are enqueued -- the data-parallel bucketer hangs
    its collective launches on it (trainer._param_groups lists the parameters in exactly this order)."""
    gpool = None
    nlev = len(blocks)
    for l in range(len(blocks) - 1, -1, -1):
        c1, c2 = blocks[l]
        lc1, lc2 = ctx.layers[l]
        g1 = dfeats[l]
        if g1 is None and gpool is None:
            raise RuntimeError("stain2stain_amd: encoder level without any incoming gradient")
        ga1 = _conv_bn_relu_bwd(c2, lc2, g1, gpool, grads, accumulate, need_dx=True)
        if on_group_done is not None:
            on_group_done()
        gpool = _conv_bn_relu_bwd(c1, lc1, ga1, None, grads, accumulate, need_dx=(l > 0), stem=(l == 0))
        if on_group_done is not None:
            on_group_done()


# ------------------------------------------------------------------------------------------------
# decoder
# ------------------------------------------------------------------------------------------------
def decoder_forward(dec, bottleneck: torch.Tensor, skips: Sequence[torch.Tensor], t_emb: torch.Tensor,
                    dtype: torch.dtype, training: bool, with_head: bool = True) -> DecCtx:
    """dec: FlowMatchingDecoder container (time_mlp, time_proj, up_blocks, outc).  with_head=False stops at the
    last activation (ctx.lows[-1]); the fused head+loss kernel of the training step takes it from there."""
    ctx = DecCtx(dtype, t_emb)
    repack_stale([cb for pair in dec.up_blocks for cb in pair], dtype)
    tbias = None
    if getattr(dec, "time_mlp", None) is not None:      # the segmentation head has no time conditioning
        l0, l2 = dec.time_mlp[0], dec.time_mlp[2]
        ctx.h1 = ops.linear_fwd(t_emb, l0.weight.detach(), l0.bias.detach())
        ctx.a1 = ops.silu_fwd(ctx.h1)
        ctx.h2 = ops.linear_fwd(ctx.a1, l2.weight.detach(), l2.bias.detach())
        tbias = ops.linear_fwd(ctx.h2, dec.time_proj.weight.detach(), dec.time_proj.bias.detach())  # [B, Cb]
    x = bottleneck
    for i, ((c1, c2), skip) in enumerate(zip(dec.up_blocks, skips)):
        B, Hs, Ws, _ = skip.shape
        up = torch.empty((B, Hs, Ws, x.shape[3]), dtype=dtype, device=x.device)
        ops.upsample2x_fwd(x, up, tbias if i == 0 else None)
        lc1, a1, _ = _conv_bn_relu(c1, skip, up, dtype, training)
        lc2, a2, _ = _conv_bn_relu(c2, a1, None, dtype, training)
        ctx.lows.append(x); ctx.ups.append(up); ctx.skips.append(skip); ctx.layers.append((lc1, lc2))
        x = a2
    ctx.lows.append(x)  # head input
    if not with_head:
        return ctx
    ctx.v = ops.head_fwd(x, dec.outc.weight.detach(), dec.outc.bias.detach() if dec.outc.bias is not None else None)
    return ctx


def decoder_backward(dec, ctx: DecCtx, dv: Optional[torch.Tensor], grads: Dict[str, torch.Tensor],
                     accumulate: bool = False, need_dt_emb: bool = False, on_group_done=None,
                     g_head: Optional[torch.Tensor] = None):
    """Returns (dbottleneck, [dskip per level], dt_emb or None); all NHWC in the compute dtype.
    ``on_group_done()``: after the head, after every conv + BatchNorm layer (last Up block first, its second conv before
    its first), then after the time path (only when the decoder has one)."""
    if g_head is not None:      # head gradients already produced by the fused head+loss kernel
        g = g_head
    else:
        g = ops.head_bwd(dv, ctx.lows[-1], dec.outc.weight.detach(), _g(grads, "outc.weight"),
                         _g(grads, "outc.bias") if dec.outc.bias is not None else None, accumulate)
    if on_group_done is not None:
        on_group_done()
    nup = len(ctx.layers)
    dskips: List[torch.Tensor] = [None] * len(ctx.layers)
    for i in range(len(ctx.layers) - 1, -1, -1):
        c1, c2 = dec.up_blocks[i]
        lc1, lc2 = ctx.layers[i]
        ga1 = _conv_bn_relu_bwd(c2, lc2, g, None, grads, accumulate, need_dx=True)
        if on_group_done is not None:
            on_group_done()
        dcat = _conv_bn_relu_bwd(c1, lc1, ga1, None, grads, accumulate, need_dx=True)
        cs = ctx.skips[i].shape[3]
        dskips[i] = dcat[..., :cs]
        low = ctx.lows[i]
        g = ops.upsample2x_bwd(dcat[..., cs:], low.shape[1], low.shape[2])
        if on_group_done is not None:
            on_group_done()
    dbott = g
    if getattr(dec, "time_mlp", None) is None:      # no time path, no parameter group for it (trainer._param_groups)
        return dbott, dskips, None
    # time path: tbias was broadcast-added to the bottleneck before the first up-sampling
    dtb = ops.pixel_sum(g)
    l0, l2 = dec.time_mlp[0], dec.time_mlp[2]
    dh2 = ops.linear_bwd(dtb, ctx.h2, dec.time_proj.weight.detach(), _g(grads, "time_proj.weight"),
                         _g(grads, "time_proj.bias"), True, accumulate)
    da1 = ops.linear_bwd(dh2, ctx.a1, l2.weight.detach(), _g(grads, "time_mlp.2.weight"),
                         _g(grads, "time_mlp.2.bias"), True, accumulate)
    dh1 = ops.silu_bwd(ctx.h1, da1)
    dt_emb = ops.linear_bwd(dh1, ctx.t_emb, l0.weight.detach(), _g(grads, "time_mlp.0.weight"),
                            _g(grads, "time_mlp.0.bias"), need_dt_emb, accumulate)
    if on_group_done is not None:
        on_group_done()
    return dbott, dskips, dt_emb
