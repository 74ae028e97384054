"""CPU oracle for the stain-translation hot path.  TEST INFRASTRUCTURE ONLY.

This is a from-scratch restatement (plain fp32 PyTorch on the host CPU) of the
conditional-flow-matching U-Net step that the reference runs through stock
``torch.nn`` layers.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product path
(``stain2stain_amd``) never does and fails loudly without its HIP library.

Parity status: PINNED.  The reference's own tests hold no vectors for this path
(SURVEY.md section 4), so the restatement is pinned against outputs of the
reference's torch-only modules run in the build container
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``), checked by
``tests/test_oracle_golden.py``.

Parameters travel as a flat ``dict[str, Tensor]`` whose keys are the reference's
``state_dict`` keys with the LightningModule attribute prefixes
(``encoder.`` / ``flow_decoder.``), e.g.
``encoder.downs.0.maxpool_conv.1.double_conv.3.weight``.

Reference lines each function follows (relative to the reference checkout):

* ``time_embedding``      src/models/components/shared_encoder.py:114-135
* ``conv_bn_relu``/``double_conv``  shared_encoder.py:12-24 (dup task_decoders.py:12-24)
* ``maxpool2``            shared_encoder.py:32-37 (nn.MaxPool2d(2))
* ``encoder_forward``     shared_encoder.py:75-104
* ``upsample2x_bilinear_ac``  task_decoders.py:34 (nn.Upsample(2, bilinear, align_corners=True))
* ``up_block``            task_decoders.py:42-50
* ``decoder_forward``     task_decoders.py:102-134
* ``flow_forward``        src/models/conditional_flow_matching_multitask.py:134-155
* ``cfm_sample``          call site conditional_flow_matching.py:66 (torchcfm 1.0.7
                          ``ConditionalFlowMatcher.sample_location_and_conditional_flow``;
                          third-party source absent -> published formula, ``t`` explicit)
* ``cfm_loss``            conditional_flow_matching.py:72
* ``adam_step``           torch.optim.Adam as configured by configs/model/*.yaml:3-7
                          via conditional_flow_matching.py:112-131
* ``euler_sample``        BASELINE.json config 4 (fixed-step Euler over the eval-mode net;
                          the reference's dopri5 lives in torchdyn, absent)
* ``seg_decoder_forward`` src/models/components/task_decoders.py:171-194 (SegmentationDecoder)
* ``dice_loss`` / ``bce_with_logits`` / ``seg_loss``
                          src/models/conditional_flow_matching_multitask.py:36-53, :119, :174-202
* ``multiclass_dice_loss`` / ``seg_loss_multiclass``
                          src/models/conditional_flow_matching_multitask_multiclassloss.py:41-83, :159, :214-245
* ``weighted_mse``        src/models/conditional_flow_matching_masked.py:76-90
* ``charbonnier_roi``     src/models/conditional_flow_matching_ROI_loss.py:78-95
* ``variant_loss_and_grads``  mask as 4th input channel (conditional_flow_matching_conditional_mask.py:62-64,
                          :73-82), optional ROI weighting of the loss, optional class row added to the time
                          embedding (build-defined stand-in for the absent third-party class-conditional U-Net)
* ``multitask_loss_and_grads``  :204-257 (flow term on xt, mask head on the source image,
                          total = flow + seg_loss_weight * seg; the encoder runs twice, so its
                          BatchNorm running statistics advance twice per step)
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------- #
# leaf operators
# --------------------------------------------------------------------------- #
def time_embedding(t: Tensor, dim: int) -> Tensor:
    """sin/cos embedding of raw t in [0,1]; frequencies exp(-i*ln(1e4)/(half-1))."""
    half = dim // 2
    step = torch.log(torch.tensor(10000.0)) / (half - 1)
    freq = torch.exp(torch.arange(half, dtype=torch.float32) * -step)
    if t.dim() == 1:
        t = t[:, None]
    ang = t.to(torch.float32) * freq[None, :]
    return torch.cat([ang.sin(), ang.cos()], dim=-1)


def batchnorm_train(x: Tensor, gamma: Tensor, beta: Tensor):
    """Per-channel batch statistics (biased variance for the normalisation).

    Returns (y, mean, biased_var).
    """
    mean = x.mean(dim=(0, 2, 3))
    var = ((x - mean[None, :, None, None]) ** 2).mean(dim=(0, 2, 3))
    inv = torch.rsqrt(var + BN_EPS)
    y = (x - mean[None, :, None, None]) * (inv * gamma)[None, :, None, None] + beta[None, :, None, None]
    return y, mean, var


def batchnorm_eval(x: Tensor, gamma: Tensor, beta: Tensor, rmean: Tensor, rvar: Tensor) -> Tensor:
    inv = torch.rsqrt(rvar + BN_EPS)
    return (x - rmean[None, :, None, None]) * (inv * gamma)[None, :, None, None] + beta[None, :, None, None]


def conv_bn_relu(x: Tensor, P: Params, conv: str, bn: str, training: bool,
                 new_buffers: Optional[Params]) -> Tensor:
    """Conv3x3(pad 1, bias) -> BatchNorm2d -> ReLU.

    In training mode the running statistics that the reference's BatchNorm2d
    would hold after this call go to ``new_buffers`` (momentum 0.1, unbiased
    variance, num_batches_tracked + 1).
    """
    z = F.conv2d(x, P[conv + ".weight"], P[conv + ".bias"], stride=1, padding=1)
    if training:
        y, mean, var = batchnorm_train(z, P[bn + ".weight"], P[bn + ".bias"])
        if new_buffers is not None:
            n = z.numel() // z.shape[1]
            unbiased = var * (n / max(n - 1, 1))
            rm = new_buffers.get(bn + ".running_mean", P[bn + ".running_mean"])
            rv = new_buffers.get(bn + ".running_var", P[bn + ".running_var"])
            nb = new_buffers.get(bn + ".num_batches_tracked", P[bn + ".num_batches_tracked"])
            new_buffers[bn + ".running_mean"] = ((1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean).detach()
            new_buffers[bn + ".running_var"] = ((1 - BN_MOMENTUM) * rv + BN_MOMENTUM * unbiased).detach()
            new_buffers[bn + ".num_batches_tracked"] = nb + 1
    else:
        y = batchnorm_eval(z, P[bn + ".weight"], P[bn + ".bias"],
                           P[bn + ".running_mean"], P[bn + ".running_var"])
    return y.clamp_min(0.0)


def double_conv(x: Tensor, P: Params, prefix: str, training: bool,
                new_buffers: Optional[Params] = None) -> Tensor:
    x = conv_bn_relu(x, P, prefix + ".double_conv.0", prefix + ".double_conv.1", training, new_buffers)
    x = conv_bn_relu(x, P, prefix + ".double_conv.3", prefix + ".double_conv.4", training, new_buffers)
    return x


def maxpool2(x: Tensor) -> Tensor:
    """2x2 / stride 2 max pooling, floor mode (odd trailing row/col dropped)."""
    b, c, h, w = x.shape
    ho, wo = h // 2, w // 2
    win = x[:, :, : 2 * ho, : 2 * wo].reshape(b, c, ho, 2, wo, 2).permute(0, 1, 2, 4, 3, 5).reshape(b, c, ho, wo, 4)
    # first maximum in (0,0),(0,1),(1,0),(1,1) order takes the value AND the gradient (ATen max_pool2d)
    idx = win.detach().argmax(dim=-1, keepdim=True)
    return win.gather(-1, idx).squeeze(-1)


def _ac_axis(n_in: int, n_out: int):
    """Source index / weights for align_corners=True linear resampling."""
    scale = torch.tensor((n_in - 1) / (n_out - 1) if n_out > 1 else 0.0, dtype=torch.float32)
    pos = torch.arange(n_out, dtype=torch.float32) * scale
    lo = pos.to(torch.int64).clamp_(max=n_in - 1)
    hi = torch.where(lo < n_in - 1, lo + 1, lo)
    w_hi = pos - lo.to(torch.float32)
    w_lo = 1.0 - w_hi
    return lo, hi, w_lo, w_hi


def upsample2x_bilinear_ac(x: Tensor) -> Tensor:
    """Bilinear x2 with align_corners=True: src = dst*(in-1)/(out-1)."""
    b, c, h, w = x.shape
    hl, hh, hwl, hwh = _ac_axis(h, 2 * h)
    wl, wh, wwl, wwh = _ac_axis(w, 2 * w)
    top = x[:, :, hl, :]
    bot = x[:, :, hh, :]
    def horiz(r: Tensor) -> Tensor:
        return r[:, :, :, wl] * wwl[None, None, None, :] + r[:, :, :, wh] * wwh[None, None, None, :]
    return horiz(top) * hwl[None, None, :, None] + horiz(bot) * hwh[None, None, :, None]


def up_block(x_low: Tensor, skip: Tensor, P: Params, prefix: str, training: bool,
             new_buffers: Optional[Params] = None) -> Tensor:
    """upsample -> zero-pad to the skip's size -> cat([skip, up]) -> DoubleConv."""
    up = upsample2x_bilinear_ac(x_low)
    dy = skip.shape[2] - up.shape[2]
    dx = skip.shape[3] - up.shape[3]
    if dy != 0 or dx != 0:
        up = F.pad(up, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
    return double_conv(torch.cat([skip, up], dim=1), P, prefix + ".conv", training, new_buffers)


def linear(x: Tensor, P: Params, name: str) -> Tensor:
    return x @ P[name + ".weight"].t() + P[name + ".bias"]


# --------------------------------------------------------------------------- #
# networks
# --------------------------------------------------------------------------- #
def n_levels(P: Params, enc: str = "encoder") -> int:
    n = 0
    while f"{enc}.downs.{n}.maxpool_conv.1.double_conv.0.weight" in P:
        n += 1
    return n


def encoder_forward(x: Tensor, P: Params, training: bool, new_buffers: Optional[Params] = None,
                    enc: str = "encoder") -> Tuple[Tensor, List[Tensor]]:
    """inc + N x (pool -> DoubleConv); returns (bottleneck, skips deepest-first)."""
    feats = [double_conv(x, P, f"{enc}.inc", training, new_buffers)]
    for i in range(n_levels(P, enc)):
        feats.append(double_conv(maxpool2(feats[-1]), P, f"{enc}.downs.{i}.maxpool_conv.1",
                                 training, new_buffers))
    return feats[-1], feats[:-1][::-1]


def decoder_forward(bottleneck: Tensor, skips: Sequence[Tensor], t_emb: Tensor, P: Params,
                    training: bool, new_buffers: Optional[Params] = None,
                    dec: str = "flow_decoder") -> Tensor:
    h = linear(t_emb, P, f"{dec}.time_mlp.0")
    h = h * torch.sigmoid(h)
    h = linear(h, P, f"{dec}.time_mlp.2")
    tb = linear(h, P, f"{dec}.time_proj")
    x = bottleneck + tb[:, :, None, None]
    for i, skip in enumerate(skips):
        x = up_block(x, skip, P, f"{dec}.ups.{i}", training, new_buffers)
    return F.conv2d(x, P[f"{dec}.outc.weight"], P[f"{dec}.outc.bias"])


def seg_decoder_forward(bottleneck: Tensor, skips: Sequence[Tensor], P: Params, training: bool,
                        new_buffers: Optional[Params] = None, dec: str = "seg_decoder") -> Tensor:
    x = bottleneck
    for i, skip in enumerate(skips):
        x = up_block(x, skip, P, f"{dec}.ups.{i}", training, new_buffers)
    return F.conv2d(x, P[f"{dec}.outc.weight"], P[f"{dec}.outc.bias"])


def dice_loss(logits: Tensor, target: Tensor, smooth: float = 1.0) -> Tensor:
    p = torch.sigmoid(logits).reshape(-1)
    g = target.reshape(-1)
    inter = (p * g).sum()
    return 1 - (2.0 * inter + smooth) / (p.sum() + g.sum() + smooth)


def bce_with_logits(logits: Tensor, target: Tensor) -> Tensor:
    z, g = logits.reshape(-1), target.reshape(-1)
    return (z.clamp_min(0) - z * g + torch.log1p(torch.exp(-z.abs()))).mean()


def seg_loss(logits: Tensor, target: Tensor, dice_weight: float = 0.5, smooth: float = 1.0):
    target = target.float()
    d, b = dice_loss(logits, target, smooth), bce_with_logits(logits, target)
    return dice_weight * d + (1 - dice_weight) * b, d, b


def multiclass_dice_loss(logits: Tensor, target: Tensor, num_classes: int, smooth: float = 1.0,
                         ignore_index: int = -100) -> Tensor:
    """conditional_flow_matching_multitask_multiclassloss.py:41-83."""
    p = torch.softmax(logits, dim=1)
    oh = F.one_hot(target.long(), num_classes).permute(0, 3, 1, 2).float()
    valid = (target != ignore_index).float().unsqueeze(1) if ignore_index >= 0 else torch.ones_like(p[:, :1])
    scores = []
    for c in range(num_classes):
        pc, gc = p[:, c:c + 1] * valid, oh[:, c:c + 1] * valid
        scores.append((2.0 * (pc * gc).sum() + smooth) / (pc.sum() + gc.sum() + smooth))
    return 1 - torch.stack(scores).mean()


def seg_loss_multiclass(logits: Tensor, target: Tensor, dice_weight: float = 0.5, ignore_index: int = -100,
                        smooth: float = 1.0):
    """:214-245 -- dw * Dice + (1-dw) * CrossEntropy(ignore_index)."""
    if target.dim() == 4 and target.shape[1] == 1:
        target = target.squeeze(1)
    target = target.long()
    d = multiclass_dice_loss(logits, target, logits.shape[1], smooth, ignore_index)
    ce = F.cross_entropy(logits, target, ignore_index=ignore_index)
    return dice_weight * d + (1 - dice_weight) * ce, d, ce


def weighted_mse(v: Tensor, u: Tensor, mask: Tensor, lam: float = 10.0) -> Tensor:
    w = (1.0 + lam * mask.float()).expand_as(v)
    return (w * (v - u) ** 2).sum() / (w.sum() + 1e-8)


def charbonnier_roi(pred: Tensor, truth: Tensor, mask: Tensor, eps_charb: float = 1e-3,
                    eps_area: float = 1e-8) -> Tensor:
    m = mask.float()
    d = pred - truth
    return (torch.sqrt(d * d + eps_charb * eps_charb) * m).sum() / (m.sum() * pred.shape[1] + eps_area)


def variant_loss_and_grads(P: Params, x0: Tensor, x1: Tensor, t: Tensor, mask: Optional[Tensor] = None,
                           mask_as_channel: bool = False, roi_lambda: Optional[float] = None,
                           y: Optional[Tensor] = None):
    """Training-mode step of the loss / conditioning variants.  Returns (loss, v, grads, new_buffers)."""
    keys = trainable_keys(P)
    Q = dict(P)
    for k in keys:
        Q[k] = P[k].detach().clone().requires_grad_(True)
    nb: Params = {}
    xt, ut = cfm_sample(x0, x1, t)
    xin = torch.cat([xt, mask.float()], dim=1) if mask_as_channel else xt
    temb = time_embedding(t, Q["flow_decoder.time_mlp.0.weight"].shape[1])
    if y is not None:
        temb = temb + Q["label_emb.weight"][y.long()]
    b, skips = encoder_forward(xin, Q, True, nb)
    v = decoder_forward(b, skips, temb, Q, True, nb)
    loss = weighted_mse(v, ut, mask, roi_lambda) if roi_lambda is not None else cfm_loss(v, ut)
    gs = torch.autograd.grad(loss, [Q[k] for k in keys])
    return loss.detach(), v.detach(), dict(zip(keys, gs)), nb


def multitask_loss_and_grads(P: Params, x0: Tensor, x1: Tensor, t: Tensor, mask: Tensor,
                             seg_loss_weight: float = 1.0, dice_weight: float = 0.5, multiclass: bool = False,
                             ignore_index: int = -100):
    """One training-mode forward/backward of the multitask step.  Returns (losses dict, grads, new_buffers)."""
    keys = trainable_keys(P)
    Q = dict(P)
    for k in keys:
        Q[k] = P[k].detach().clone().requires_grad_(True)
    nb: Params = {}
    xt, ut = cfm_sample(x0, x1, t)
    flow = cfm_loss(flow_forward(t, xt, Q, True, nb), ut)
    b, skips = encoder_forward(x0, Q, True, nb)
    logits = seg_decoder_forward(b, skips, Q, True, nb)
    seg, d, bce = (seg_loss_multiclass(logits, mask, dice_weight, ignore_index) if multiclass
                   else seg_loss(logits, mask, dice_weight))
    total = flow + seg_loss_weight * seg
    gs = torch.autograd.grad(total, [Q[k] for k in keys])
    losses = {"total": total.detach(), "flow": flow.detach(), "seg": seg.detach(), "dice": d.detach(),
              "bce": bce.detach(), "ce": bce.detach()}
    return losses, dict(zip(keys, gs)), nb


def flow_forward(t: Tensor, x: Tensor, P: Params, training: bool,
                 new_buffers: Optional[Params] = None, time_emb_dim: Optional[int] = None) -> Tensor:
    """v = decoder(encoder(x), TimeEmbedding(t)); scalar t is broadcast over the batch."""
    if time_emb_dim is None:
        time_emb_dim = P["flow_decoder.time_mlp.0.weight"].shape[1]
    if t.dim() == 0:
        t = t[None].expand(x.shape[0])
    elif t.dim() == 1 and t.shape[0] == 1:
        t = t.expand(x.shape[0])
    b, skips = encoder_forward(x, P, training, new_buffers)
    return decoder_forward(b, skips, time_embedding(t, time_emb_dim), P, training, new_buffers)


# --------------------------------------------------------------------------- #
# flow matching, optimiser, sampler
# --------------------------------------------------------------------------- #
def cfm_sample(x0: Tensor, x1: Tensor, t: Tensor, sigma: float = 0.0,
               eps: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """xt = t*x1 + (1-t)*x0 + sigma*eps ; ut = x1 - x0  (t given explicitly)."""
    tb = t.reshape(-1, 1, 1, 1)
    xt = tb * x1 + (1.0 - tb) * x0
    if sigma != 0.0:
        xt = xt + sigma * (eps if eps is not None else torch.randn_like(x0))
    return xt, x1 - x0


def cfm_loss(v: Tensor, ut: Tensor) -> Tensor:
    return ((v - ut) ** 2).mean()


def trainable_keys(P: Params) -> List[str]:
    skip = ("running_mean", "running_var", "num_batches_tracked")
    return [k for k in P if not k.endswith(skip)]


def loss_and_grads(P: Params, x0: Tensor, x1: Tensor, t: Tensor):
    """One training-mode forward/backward.  Returns (loss, v, grads, new_buffers)."""
    keys = trainable_keys(P)
    Q = dict(P)
    for k in keys:
        Q[k] = P[k].detach().clone().requires_grad_(True)
    new_buffers: Params = {}
    xt, ut = cfm_sample(x0, x1, t)
    v = flow_forward(t, xt, Q, True, new_buffers)
    loss = cfm_loss(v, ut)
    gs = torch.autograd.grad(loss, [Q[k] for k in keys])
    return loss.detach(), v.detach(), dict(zip(keys, gs)), new_buffers


def adam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float = 1e-4,
              beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8,
              weight_decay: float = 0.0):
    """torch.optim.Adam (L2 weight decay folded into the gradient, no amsgrad)."""
    if weight_decay != 0.0:
        g = g + weight_decay * p
    m = m + (g - m) * (1.0 - beta1)
    v = v * beta2 + (1.0 - beta2) * g * g
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


def train_steps(P: Params, batches, lr: float = 1e-4, weight_decay: float = 1e-5):
    """Run len(batches) optimisation steps; batches = [(x0, x1, t), ...].

    Returns (final params incl. BN buffers, list of per-step dicts).
    """
    P = {k: v.clone() for k, v in P.items()}
    keys = trainable_keys(P)
    m = {k: torch.zeros_like(P[k]) for k in keys}
    v2 = {k: torch.zeros_like(P[k]) for k in keys}
    hist = []
    for step, (x0, x1, t) in enumerate(batches, start=1):
        loss, v, grads, nb = loss_and_grads(P, x0, x1, t)
        for k in keys:
            P[k], m[k], v2[k] = adam_step(P[k], grads[k], m[k], v2[k], step, lr=lr,
                                          weight_decay=weight_decay)
        P.update(nb)
        hist.append({"loss": loss, "v": v, "grads": grads})
    return P, hist


@torch.no_grad()
def euler_sample(P: Params, x: Tensor, n_steps: int = 50) -> Tensor:
    """x <- x + (1/n) * v(t_k, x), t_k = k/n, eval-mode network (BN running stats)."""
    dt = 1.0 / n_steps
    for k in range(n_steps):
        t = torch.full((x.shape[0],), k * dt, dtype=torch.float32)
        x = x + dt * flow_forward(t, x, P, False)
    return x
