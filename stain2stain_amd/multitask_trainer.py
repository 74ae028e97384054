"""One optimisation step of the multitask model (BASELINE.json configs[4]) without autograd:

    t, xt, ut   <- probability path sample                       (conditional_flow_matching_multitask.py:224)
    flow loss   <- mean((flow_decoder(encoder(xt), t) - ut)^2)   (:228-231)
    seg loss    <- dw Dice + (1 - dw) BCE on seg_decoder(encoder(source))   (:236-246; softmax Dice + CE with num_classes)
    total       <- flow + seg_loss_weight * seg                   (:249)
    grads       <- backward through both heads and BOTH encoder passes (the encoder's gradients add up)
    params      <- one Adam over encoder + flow_decoder + seg_decoder (:391-417)

The shared encoder runs twice per step (on xt, then on the source image: BatchNorm's running statistics are updated in
that order, as in the reference), so this step is 668 GFLOP per 256x256 tile against the plain flow step's 334 -- "the
best showcase for conv throughput" (SURVEY 8f).  The module path (``MultiTaskFlowMatchingModule`` under autograd +
any torch optimiser) computes the same numbers; this class is its fused form: the engine's passes directly, the flow
head + MSE in one kernel, weight gradients on a side HIP stream, parameters / gradients / Adam moments in flat fp32
buffers in backward-completion order (gradient buckets are contiguous slices), one fused Adam, one batched repack.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from . import engine, ops
from .components import FlowMatchingDecoder, SegmentationDecoder, SharedEncoder
from .ddp import all_reduce_mean_scalar
from .pix2pix_engine import FlatParams


def _layer_groups(named: Dict[str, torch.nn.Parameter], prefix: str, owner: str) -> List[List[Tuple[str, torch.nn.Parameter]]]:
    """The two conv + BatchNorm layers of a DoubleConv in the order their backward finishes: (3, 4) before (0, 1)."""
    out = []
    for idx in (("3.", "4."), ("0.", "1.")):
        out.append([(f"{owner}.{n}", p) for n, p in named.items() if any(n.startswith(prefix + i) for i in idx)])
    return out


def _decoder_groups(dec, owner: str) -> List[List[Tuple[str, torch.nn.Parameter]]]:
    """engine.decoder_backward's on_group_done order: head, Up blocks from the last, then the time path (if any)."""
    named = dict(dec.named_parameters())
    groups = [[(f"{owner}.{n}", p) for n, p in named.items() if n.startswith("outc.")]]
    for i in range(len(dec.ups) - 1, -1, -1):
        groups += _layer_groups(named, f"ups.{i}.conv.double_conv.", owner)
    if getattr(dec, "time_mlp", None) is not None:
        groups.append([(f"{owner}.{n}", p) for n, p in named.items() if n.startswith("time_")])
    return groups


def _encoder_groups(enc, owner: str) -> List[List[Tuple[str, torch.nn.Parameter]]]:
    named = dict(enc.named_parameters())
    groups = []
    for i in range(len(enc.downs) - 1, -1, -1):
        groups += _layer_groups(named, f"downs.{i}.maxpool_conv.1.double_conv.", owner)
    return groups + _layer_groups(named, "inc.double_conv.", owner)


class MultiTaskTrainer:
    """Fused training step for ``SharedEncoder`` + ``FlowMatchingDecoder`` + ``SegmentationDecoder`` (their parameters
    become views of this trainer's flat buffers; ``state_dict`` / ``load_state_dict`` of the modules keep working).

    ``num_classes`` None: binary mask head, Dice + BCE; an integer: softmax Dice + CrossEntropy on class-index masks
    (conditional_flow_matching_multitask_multiclassloss.py:92-159), ``ignore_index`` as there."""

    def __init__(self, encoder: SharedEncoder, flow_decoder: FlowMatchingDecoder, seg_decoder: SegmentationDecoder,
                 time_emb_dim: int = 256, lr: float = 1e-4, weight_decay: float = 1e-5,
                 betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8, seg_loss_weight: float = 1.0,
                 dice_weight: float = 0.5, dice_smooth: float = 1.0, num_classes: Optional[int] = None,
                 ignore_index: int = -100, sigma: float = 0.0, bucket_mb: float = 4.0, process_group=None,
                 sync_loss: bool = True, max_bucket_mb: float = 16.0, sharded_optimizer: bool = False):
        dev = next(encoder.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("stain2stain_amd: MultiTaskTrainer needs the networks on a GPU (HIP-only implementation)")
        if not (encoder.compute_dtype == flow_decoder.compute_dtype == seg_decoder.compute_dtype):
            raise ValueError("encoder and decoders must share one precision")
        self.encoder, self.flow_decoder, self.seg_decoder = encoder, flow_decoder, seg_decoder
        self.time_emb_dim, self.sigma = time_emb_dim, sigma
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, tuple(betas), eps
        self.seg_loss_weight, self.dice_weight, self.dice_smooth = seg_loss_weight, dice_weight, dice_smooth
        self.num_classes, self.ignore_index = num_classes, ignore_index
        self.pg, self.sync_loss = process_group, sync_loss
        self._dtype = encoder.compute_dtype
        # backward-completion order: mask head and its Up blocks, flow head, its Up blocks and time MLP, then the encoder
        # (whose gradients are complete only when the SECOND of its two backward passes has run)
        groups = _decoder_groups(seg_decoder, "seg_decoder") + _decoder_groups(flow_decoder, "flow_decoder") \
            + _encoder_groups(encoder, "encoder")
        n_named = sum(1 for m in (encoder, flow_decoder, seg_decoder) for _ in m.parameters())
        if sum(len(g) for g in groups) != n_named or any(not g for g in groups):
            raise RuntimeError("parameter grouping does not cover the networks")
        self.fp = FlatParams(groups, bucket_mb, process_group, max_bucket_mb, sharded_optimizer)
        gr = self.fp.grads
        self.grads_enc = {k[len("encoder."):]: v for k, v in gr.items() if k.startswith("encoder.")}
        self.grads_flow = {k[len("flow_decoder."):]: v for k, v in gr.items() if k.startswith("flow_decoder.")}
        self.grads_seg = {k[len("seg_decoder."):]: v for k, v in gr.items() if k.startswith("seg_decoder.")}
        from .ddp import broadcast_from_rank0
        broadcast_from_rank0([b for m in (encoder, flow_decoder, seg_decoder) for b in m.buffers()], process_group)
        # one launch re-packs every MFMA conv's weights (in fp32 mode all but the stem, which reads the master directly)
        blocks = list(encoder._blocks) + list(flow_decoder.up_blocks) + list(seg_decoder.up_blocks)
        self._packed = [cb for pair in blocks for cb in pair if cb is not encoder._blocks[0][0] or
                        (self._dtype == torch.bfloat16 and cb.cin <= 8)]     # (the bf16 stem runs as a packed conv too)
        rows, start = [], 0
        for cb in self._packed:
            wf, wd = cb.ensure_buffers(self._dtype)
            rows.append([cb.conv.weight.data_ptr(), wf.data_ptr(), wd.data_ptr(), cb.cout, cb.cin, start])
            start += ((cb.cout + 31) // 32) * ((cb.cin + 31) // 32)
        self._pack_desc = torch.tensor(rows, dtype=torch.int64, device=dev)
        self._pack_total = start
        self._side = ops.side_stream_for(dev)
        self.overlap_wgrad = True
        self._group = 0
        self._repack()

    @property
    def step_count(self) -> int:
        return self.fp.step_count

    def _repack(self) -> None:
        ops.pack_conv3x3_batched(self._pack_desc, self._pack_total, self._dtype)
        for cb in self._packed:
            cb.mark_packed(self._dtype)

    def _group_done(self) -> None:
        self.fp.bucketer.mark_ready_ordered(self._group, self._side if self.overlap_wgrad else None)
        self._group += 1

    # ------------------------------------------------------------------------------------------
    def forward_backward(self, source: torch.Tensor, target: torch.Tensor, mask: torch.Tensor,
                         t: Optional[torch.Tensor] = None, want_outputs: bool = False):
        """Gradients of the local batch into the flat buffer (+ async exchange).  Returns (losses, outputs): ``losses`` =
        {"total", "flow", "seg", "seg_dice", "seg_bce" | "seg_ce"} as device scalars (the reference's logging keys,
        conditional_flow_matching_multitask.py:251-257); ``outputs`` = {"v", "logits"} (NCHW fp32) when asked."""
        enc, fdec, sdec = self.encoder, self.flow_decoder, self.seg_decoder
        dt = self._dtype
        x0, x1 = source.float().contiguous(), target.float().contiguous()
        B = x0.shape[0]
        if t is None:
            t = torch.rand(B, device=x0.device, dtype=torch.float32)
        t = t.float().contiguous()
        eps_noise = torch.randn_like(x0) if self.sigma != 0.0 else None
        xt, ut = ops.cfm_sample(x0, x1, t, self.sigma, eps_noise)
        # forward: encoder(xt) -> flow head, then encoder(source) -> mask head (the reference's order, :228-236)
        ectx1 = engine.encoder_forward(enc._blocks, xt, dt, True)
        temb = ops.time_embedding(t, self.time_emb_dim)
        f1 = ectx1.feats
        dctx_f = engine.decoder_forward(fdec, f1[-1], f1[:-1][::-1], temb, dt, True, with_head=False)
        ectx2 = engine.encoder_forward(enc._blocks, x0, dt, True)
        f2 = ectx2.feats
        dctx_s = engine.decoder_forward(sdec, f2[-1], f2[:-1][::-1], None, dt, True)
        logits = dctx_s.v
        flow, g_head, v = ops.head_loss_fused(dctx_f.lows[-1], fdec.outc.weight.detach(),
                                              fdec.outc.bias.detach() if fdec.outc.bias is not None else None, ut,
                                              self.grads_flow["outc.weight"], self.grads_flow.get("outc.bias"),
                                              want_v=want_outputs)
        w = float(self.seg_loss_weight)
        if self.num_classes is not None:
            tgt = mask[:, 0] if (mask.dim() == 4 and mask.shape[1] == 1) else mask
            seg3, dz = ops.seg_loss_multiclass(logits, tgt.long(), self.ignore_index, self.dice_smooth, self.dice_weight,
                                               want_grad=True, grad_scale=w, validate=False)
        else:
            seg3, dz = ops.seg_loss(logits, mask.float().reshape(logits.shape), self.dice_smooth, self.dice_weight,
                                    want_grad=True, grad_scale=w)
        total = flow + w * seg3[0]                                       # (two device scalars: the logged total, :249)
        self.fp.bucketer.start_step()
        self._group = 0
        engine.side_stream = self._side if self.overlap_wgrad else None
        try:
            L = len(f1) - 1
            dbs, dss, _ = engine.decoder_backward(sdec, dctx_s, dz, self.grads_seg, on_group_done=self._group_done)
            dbf, dsf, _ = engine.decoder_backward(fdec, dctx_f, None, self.grads_flow, g_head=g_head,
                                                  on_group_done=self._group_done)
            # the encoder's two passes: the source pass writes its gradients, the xt pass adds to them and closes the groups
            engine.encoder_backward(enc._blocks, ectx2, [dss[L - 1 - l] for l in range(L)] + [dbs], self.grads_enc)
            engine.encoder_backward(enc._blocks, ectx1, [dsf[L - 1 - l] for l in range(L)] + [dbf], self.grads_enc,
                                    accumulate=True, on_group_done=self._group_done)
        finally:
            engine.side_stream = None
        engine.join_side(self._side)
        second = "seg_ce" if self.num_classes is not None else "seg_bce"
        losses = {"total": total, "flow": flow, "seg": seg3[0], "seg_dice": seg3[1], second: seg3[2]}
        return losses, ({"v": v, "logits": logits} if want_outputs else None)

    def optimizer_step(self) -> None:
        engine.join_side(self._side)
        self.fp.adam(self.lr, self.betas, self.eps, self.wd)
        self._repack()
        engine.mutation_epoch[0] += 1

    def step(self, source: torch.Tensor, target: torch.Tensor, mask: torch.Tensor,
             t: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One training step on this rank's shard of the global batch; returns the (rank-mean) total loss."""
        losses, _ = self.forward_backward(source, target, mask, t)
        total = losses["total"]
        work = all_reduce_mean_scalar(total, self.pg) if self.sync_loss else None
        self.optimizer_step()
        if work is not None:
            work.wait()
            total = total / dist.get_world_size(self.pg)
        return total
