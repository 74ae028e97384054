"""One whole optimisation step of the stain-translation model without autograd:

    t, xt, ut  <- probability path sample        (conditional_flow_matching.py:66)
    v          <- U-Net(t, xt)                    (:69)
    loss       <- mean((v-ut)^2)                  (:72)
    grads      <- backward through the U-Net      (Lightning automatic optimisation)
    grads      <- bucketed RCCL all-reduce, overlapped with the rest of the backward (DDP, C1)
    params     <- Adam                            (configure_optimizers, :112-131)

This is the path ``bench.py`` times.  The LightningModule-compatible path (autograd Functions in
components.py + any torch optimiser) produces the same numbers; this class removes the autograd /
optimiser Python overhead, keeps all parameters, gradients and Adam moments in three flat fp32
buffers and overlaps communication with compute.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import os

import torch
import torch.distributed as dist

from . import engine, ops
from .components import FlowUNet
from .ddp import ALIGN, GradBucketer, all_reduce_mean_scalar, broadcast_from_rank0


def _param_groups(net: FlowUNet) -> List[List[Tuple[str, str, torch.nn.Parameter]]]:
    """Parameters grouped in the order the backward pass finishes them, one group per conv + BatchNorm LAYER
    (engine.decoder_backward / encoder_backward fire ``on_group_done`` in exactly this order): the head, every Up
    block from the last (its second conv before its first), the time MLP, every encoder level from the deepest.

    Each entry is (owner, local_name, parameter) with owner in {"dec", "enc"}.
    """
    dec, enc = net.flow_decoder, net.encoder
    dnamed = dict(dec.named_parameters())
    enamed = dict(enc.named_parameters())
    groups: List[List[Tuple[str, str, torch.nn.Parameter]]] = []

    def layer(owner, named, prefix):        # the two layers of a DoubleConv: (conv 3, bn 4) finishes before (conv 0, bn 1)
        for idx in (("3.", "4."), ("0.", "1.")):
            groups.append([(owner, n, p) for n, p in named.items()
                           if any(n.startswith(prefix + i) for i in idx)])

    groups.append([("dec", n, p) for n, p in dnamed.items() if n.startswith("outc.")])
    for i in range(len(dec.ups) - 1, -1, -1):
        layer("dec", dnamed, f"ups.{i}.conv.double_conv.")
    groups.append([("dec", n, p) for n, p in dnamed.items() if n.startswith("time_")])
    for i in range(len(enc.downs) - 1, -1, -1):
        layer("enc", enamed, f"downs.{i}.maxpool_conv.1.double_conv.")
    layer("enc", enamed, "inc.double_conv.")
    groups = [g for g in groups if g] if getattr(dec, "time_mlp", None) is None else groups
    seen = sum(len(g) for g in groups)
    if seen != len(dnamed) + len(enamed) or any(not g for g in groups):
        raise RuntimeError("parameter grouping does not cover the network")
    return groups


class FusedAdamHandle(torch.optim.Optimizer):
    """A ``torch.optim.Optimizer`` face for the trainer's fused Adam, so that the reference's schedulers
    (``ReduceLROnPlateau`` in configs/model/*.yaml:12-16, driven by Lightning on ``val/loss``) and anything else that
    edits ``optimizer.param_groups[0]["lr"]`` work unchanged: the trainer reads its hyper-parameters from this
    param group at every step.  ``step()`` runs the trainer's optimiser step (join the all-reduce, fused Adam,
    repack)."""

    def __init__(self, trainer: "CFMTrainer", lr, betas, eps, weight_decay):
        self._trainer = trainer
        super().__init__([trainer.flat_p], dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        self._trainer.optimizer_step()
        return loss

    def zero_grad(self, set_to_none: bool = False) -> None:      # every backward overwrites the flat gradient buffer
        pass

    # The handle's own parameter list is the one flat buffer, which carries no per-parameter state: a checkpoint
    # written from the inherited state_dict() would silently drop exp_avg / exp_avg_sq / step.  Lightning and user
    # code checkpoint optimisers through these two methods, so they speak torch.optim.Adam's layout over
    # net.parameters() (what the reference's checkpoint["optimizer_states"][0] holds).
    def state_dict(self):
        return self._trainer.optimizer_state_dict()

    def load_state_dict(self, state_dict) -> None:
        self._trainer.load_optimizer_state_dict(state_dict)


class CFMTrainer:
    def __init__(self, net: FlowUNet, lr: float = 1e-4, weight_decay: float = 1e-5,
                 betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8, sigma: float = 0.0,
                 bucket_mb: float = 4.0, process_group=None, sync_loss: bool = True, max_bucket_mb: float = 16.0,
                 sharded_optimizer: bool = False, sync_batchnorm: bool = False, graph: bool = False):
        """``graph``: ``step()`` replays ONE captured hipGraph per optimisation step (sample -> forward -> loss -> backward
        -> [gradient exchange] -> Adam -> repack: ~210 kernel launches, the side-stream fork / joins included) instead of
        issuing them from Python; the first step of a batch shape runs eagerly (it is also the warm-up), the second
        captures.  Same kernels in the same order: bit-equal to the eager step.  The host only refreshes the eight Adam
        scalars (step count -> bias corrections, scheduler -> lr) in device memory before each replay.
        ``sharded_optimizer``: exchange gradients by reduce-scatter, run the fused Adam on this rank's 1/world of every
        bucket and all-gather the updated parameters (ddp.GradBucketer, mode "reduce_scatter") instead of all-reduce +
        a full Adam pass on every rank.  Same results, same wire volume, 1/world of the optimiser's HBM traffic.
        ``sync_batchnorm``: BatchNorm statistics over the global batch (Lightning's ``sync_batchnorm: True``,
        configs/trainer/ddp.yaml:9; the stain experiments leave it off): two 2C-float all-reduces per layer and step."""
        dev = next(net.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("stain2stain_amd: CFMTrainer needs the network on a GPU (HIP-only implementation)")
        self.net = net
        self.sigma = sigma
        self.pg = process_group
        self.sync_loss = sync_loss
        self.step_count = 0
        groups = _param_groups(net)
        # 8-float alignment of every parameter keeps the flat views 32-byte aligned; groups (= layers) start on
        # multiples of ddp.ALIGN floats so that a bucket divides into 32-byte-aligned shards for up to 8 ranks
        sizes, offs, off = [], {}, 0
        for g in groups:
            start = off
            for owner, name, p in g:
                offs[(owner, name)] = off
                off += (p.numel() + 7) // 8 * 8
            off = (off + ALIGN - 1) // ALIGN * ALIGN
            sizes.append(off - start)
        self.flat_p = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grads_enc: Dict[str, torch.Tensor] = {}
        self.grads_dec: Dict[str, torch.Tensor] = {}
        with torch.no_grad():
            for g in groups:
                for owner, name, p in g:
                    o, n = offs[(owner, name)], p.numel()
                    self.flat_p[o:o + n].copy_(p.detach().reshape(-1))
                    p.data = self.flat_p[o:o + n].view(p.shape)
                    gv = self.flat_g[o:o + n].view(p.shape)
                    p.grad = gv
                    (self.grads_dec if owner == "dec" else self.grads_enc)[name] = gv
        self._slot = {id(p): (offs[(owner, name)], p.numel()) for g in groups for owner, name, p in g}
        self.optimizer = FusedAdamHandle(self, lr, betas, eps, weight_decay)
        self.n_params = sum(p.numel() for g in groups for _, _, p in g)
        self.bucketer = GradBucketer(self.flat_g, sizes, bucket_mb, process_group, max_bucket_mb,
                                     "reduce_scatter" if sharded_optimizer else "allreduce")
        self._group = 0
        self._sync_bn = engine.SyncBNExchange(process_group) if sync_batchnorm and self.bucketer.enabled else None
        broadcast_from_rank0([self.flat_p] + [b for b in net.buffers()], process_group)
        self._blocks = list(net.encoder._blocks) + list(net.flow_decoder.up_blocks)
        # one launch re-packs every MFMA conv's weights (in fp32 mode all but the stem, which reads the master directly)
        self._dtype = net.encoder.compute_dtype
        self._packed = [cb for pair in self._blocks for cb in pair if cb is not net.encoder._blocks[0][0] or
                        (self._dtype == torch.bfloat16 and cb.cin <= 8)]     # (the bf16 stem runs as a packed conv too)
        rows, start = [], 0
        for cb in self._packed:
            wf, wd = cb.ensure_buffers(self._dtype)
            rows.append([cb.conv.weight.data_ptr(), wf.data_ptr(), wd.data_ptr(), cb.cout, cb.cin, start])
            start += ((cb.cout + 31) // 32) * ((cb.cin + 31) // 32)
        # weight gradients overlap the bandwidth-bound backward passes on a second HIP stream (engine.run_on_side);
        # S2S_WGRAD_STREAM=0 keeps everything on one stream
        self._side = ops.side_stream_for(dev)
        self.overlap_wgrad = True       # bench.py clears it on the steps whose kernels it brackets with HIP events
        self._pack_desc = torch.tensor(rows, dtype=torch.int64, device=dev)
        self._pack_total = start
        self._repack()
        self.graph = graph
        self._captured = None                  # _CapturedStep of the current batch shape
        self._warm_key = None
        self._hyper = None                     # ops.AdamHyperRing, created with the first captured step

    # hyper-parameters live in the optimiser handle's param group (schedulers edit them there)
    @property
    def _hp(self):
        return self.optimizer.param_groups[0]

    lr = property(lambda self: self._hp["lr"], lambda self, v: self._hp.__setitem__("lr", v))
    wd = property(lambda self: self._hp["weight_decay"], lambda self, v: self._hp.__setitem__("weight_decay", v))
    betas = property(lambda self: self._hp["betas"], lambda self, v: self._hp.__setitem__("betas", tuple(v)))
    eps = property(lambda self: self._hp["eps"], lambda self, v: self._hp.__setitem__("eps", v))

    # ------------------------------------------------------------------------------------------
    def forward_backward(self, x0: torch.Tensor, x1: torch.Tensor, t: Optional[torch.Tensor] = None,
                         want_v: bool = True):
        """Gradients of the local batch into the flat buffer (+ async all-reduce).  Returns (loss, v); the
        velocity field is only materialised when ``want_v`` (tests) -- the training step does not need it."""
        net = self.net
        enc, dec = net.encoder, net.flow_decoder
        dt = enc.compute_dtype
        B = x0.shape[0]
        x0, x1 = x0.float().contiguous(), x1.float().contiguous()       # NCHW fp32, the reference's boundary layout
        if t is None:
            t = torch.rand(B, device=x0.device, dtype=torch.float32)
        t = t.float().contiguous()
        eps_noise = torch.randn_like(x0) if self.sigma != 0.0 else None
        xt, ut = ops.cfm_sample(x0, x1, t, self.sigma, eps_noise)
        with engine.sync_batchnorm(self._sync_bn):
            ectx = engine.encoder_forward(enc._blocks, xt, dt, True)
            temb = ops.time_embedding(t, net.time_embedding.dim)
            feats = ectx.feats
            dctx = engine.decoder_forward(dec, feats[-1], feats[:-1][::-1], temb, dt, True, with_head=False)
        # head conv + loss + their backward in one sweep over the last activation
        loss, g_head, v = ops.head_loss_fused(dctx.lows[-1], dec.outc.weight.detach(),
                                              dec.outc.bias.detach() if dec.outc.bias is not None else None, ut,
                                              self.grads_dec["outc.weight"], self.grads_dec.get("outc.bias"),
                                              want_v=want_v)
        dctx.v = v
        self.bucketer.start_step()
        self._group = 0
        engine.side_stream = self._side if self.overlap_wgrad else None
        try:
            dbott, dskips, _ = engine.decoder_backward(dec, dctx, None, self.grads_dec, g_head=g_head,
                                                       on_group_done=self._group_done)
            L = len(feats) - 1
            dfeats = [dskips[L - 1 - l] for l in range(L)] + [dbott]
            engine.encoder_backward(enc._blocks, ectx, dfeats, self.grads_enc, on_group_done=self._group_done)
        finally:
            engine.side_stream = None
        engine.join_side(self._side)        # whatever reads the gradients next (Adam, a test) is on the compute stream
        return loss, dctx.v

    def _group_done(self) -> None:
        """The next parameter group (layer, in _param_groups order) has its gradients enqueued: BatchNorm / bias / linear
        gradients on the compute stream, the conv weight gradient on the side stream.  A bucket that closes here is
        exchanged behind BOTH streams (GradBucketer.mark_ready_ordered): the head and the time-MLP groups have no
        side-stream fork after their compute-stream kernels, so ordering behind the side stream alone is not enough."""
        self.bucketer.mark_ready_ordered(self._group, self._side if self.overlap_wgrad else None)
        self._group += 1

    def optimizer_step(self) -> None:
        self.step_count += 1
        self._enqueue_optimizer(None)
        engine.mutation_epoch[0] += 1

    def _enqueue_optimizer(self, hyper_dev: Optional[torch.Tensor]) -> None:
        """Join the exchange, Adam, [parameter all-gather], repack.  ``hyper_dev``: the Adam scalars in device memory
        (captured step) instead of kernel arguments."""
        engine.join_side(self._side)        # the weight gradients
        self.bucketer.wait_all()
        for lo, hi in self.bucketer.shards():          # everything, or this rank's slices in sharded mode
            if hyper_dev is None:
                ops.adam_step_(self.flat_p[lo:hi], self.flat_g[lo:hi], self.flat_m[lo:hi], self.flat_v[lo:hi],
                               self.step_count, self.lr, self.betas[0], self.betas[1], self.eps, self.wd,
                               self.bucketer.grad_scale)
            else:
                ops.adam_step_dev_(self.flat_p[lo:hi], self.flat_g[lo:hi], self.flat_m[lo:hi], self.flat_v[lo:hi], hyper_dev)
        self.bucketer.all_gather(self.flat_p)
        self._repack()              # master weights changed behind torch's version counter

    # ------------------------------------------------------------------------------------------
    # optimiser state in torch.optim.Adam's own layout, so a run can move between this trainer and the
    # reference's Lightning loop (checkpoint["optimizer_states"][0], parameters in net.parameters() order)
    def optimizer_state_dict(self) -> Dict:
        """With ``sharded_optimizer`` a rank's Adam moments are current only on its own 1/world of every bucket: they are
        all-gathered first, so the dump is the full state on every rank (a COLLECTIVE in that mode: every rank calls it,
        as Lightning's checkpoint hook does)."""
        if self.bucketer.mode == "reduce_scatter" and self.bucketer.enabled and self.step_count > 0:
            self.bucketer.all_gather(self.flat_m)
            self.bucketer.all_gather(self.flat_v)
        state = {}
        for i, p in enumerate(self.net.parameters()):
            o, n = self._slot[id(p)]
            if self.step_count > 0:
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.flat_m[o:o + n].view(p.shape).clone(),
                            "exp_avg_sq": self.flat_v[o:o + n].view(p.shape).clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.wd,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                 "fused": None, "params": list(range(len(self._slot)))}
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, sd: Dict) -> None:
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self._slot):
            raise ValueError("optimizer state does not match this network (one group over net.parameters() expected)")
        if groups[0].get("amsgrad", False) or groups[0].get("maximize", False):
            raise ValueError("amsgrad / maximize Adam states are not supported")
        g = groups[0]
        self.lr, self.betas, self.eps, self.wd = g["lr"], tuple(g["betas"]), g["eps"], g["weight_decay"]
        steps = set()
        self.flat_m.zero_()
        self.flat_v.zero_()
        for i, p in enumerate(self.net.parameters()):
            st = sd["state"].get(i, sd["state"].get(str(i)))
            if st is None:
                continue
            o, n = self._slot[id(p)]
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError(f"optimizer state {i}: shape {tuple(st['exp_avg'].shape)} != {tuple(p.shape)}")
            self.flat_m[o:o + n].copy_(st["exp_avg"].reshape(-1))
            self.flat_v[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("per-parameter step counts differ; the flat Adam kernel keeps one")
        self.step_count = steps.pop() if steps else 0

    def _repack(self) -> None:
        ops.pack_conv3x3_batched(self._pack_desc, self._pack_total, self._dtype)
        for cb in self._packed:
            cb.mark_packed(self._dtype)

    def step(self, x0: torch.Tensor, x1: torch.Tensor, t: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One training step on this rank's shard of the global batch; returns the (rank-mean) loss."""
        with ops.workspace_owner(self):          # split-K / weight-gradient slabs belong to this trainer (ops._workspace)
            if self.graph and ops._PROFILE is None:
                return self._step_graphed(x0, x1, t)
            self.step_count += 1
            loss = self._step_body(x0, x1, t, None)
            engine.mutation_epoch[0] += 1
            return loss

    def _step_body(self, x0, x1, t, hyper_dev) -> torch.Tensor:
        loss, _ = self.forward_backward(x0, x1, t, want_v=False)
        work = all_reduce_mean_scalar(loss, self.pg) if self.sync_loss else None
        self._enqueue_optimizer(hyper_dev)
        if work is not None:
            work.wait()
            loss = loss / dist.get_world_size(self.pg)
        return loss

    def _step_graphed(self, x0: torch.Tensor, x1: torch.Tensor, t: Optional[torch.Tensor]) -> torch.Tensor:
        B = x0.shape[0]
        if t is None:
            t = torch.rand(B, device=x0.device, dtype=torch.float32)
        if self.sigma != 0.0:
            raise NotImplementedError("graph=True with sigma != 0: the path noise is drawn inside forward_backward")
        if self.bucketer.enabled and dist.get_backend(self.pg) != "nccl":
            raise RuntimeError("stain2stain_amd: graph=True captures the gradient exchange; only RCCL ('nccl') "
                               "collectives can be captured")
        if self._hyper is None:
            self._hyper = ops.AdamHyperRing(x0.device)
        key = (tuple(x0.shape), x0.device, self.overlap_wgrad)
        if self.bucketer.enabled and not CFMTrainer._warned_multi_rank_graph:
            CFMTrainer._warned_multi_rank_graph = True
            import warnings
            warnings.warn("stain2stain_amd: graph=True with a gradient exchange captures the RCCL collectives; this has "
                          "only been exercised at world size 1 (S2S_FORCE_DDP) -- treat it as experimental", stacklevel=3)
        # the step counter moves only once the step has run (a failed capture must not count as a step)
        step = self.step_count + 1
        if self._captured is None or self._captured.key != key:
            if self._warm_key != key:       # first step of this shape: eager (module load, LDS attributes, workspaces)
                self._warm_key, self._captured = key, None
                self.step_count = step
                try:
                    loss = self._step_body(x0, x1, t, None)
                except BaseException:
                    self.step_count = step - 1
                    raise
                engine.mutation_epoch[0] += 1
                return loss
            cap = _CapturedStep(key, x0, x1, t)
            torch.cuda.synchronize()
            with ops.capture_graph(cap.graph):
                cap.loss = self._step_body(cap.x0, cap.x1, cap.t, self._hyper.dev)
            self._captured = cap
        cap = self._captured
        cap.x0.copy_(x0); cap.x1.copy_(x1); cap.t.copy_(t)
        self._hyper.push(ops.adam_hyper(step, self.lr, self.betas[0], self.betas[1], self.eps, self.wd,
                                        self.bucketer.grad_scale))
        cap.graph.replay()
        self.step_count = step
        engine.mutation_epoch[0] += 1
        return cap.loss.clone()

    _warned_multi_rank_graph = False

    def close(self) -> None:
        """Drop the captured graph, its static buffers and the pinned Adam-scalar ring while torch and the HIP runtime are
        fully alive (a captured step sits in a reference cycle with its trainer; left to the cyclic collector it may be
        destroyed during another capture or at interpreter exit, in an order the runtime does not define)."""
        self._captured, self._warm_key, self._hyper = None, None, None
        ops.release_workspaces(self)

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class _CapturedStep:
    """Static input buffers + the hipGraph of one optimisation step for one batch shape."""

    def __init__(self, key, x0, x1, t):
        self.key = key
        self.x0 = torch.empty_like(x0, dtype=torch.float32).contiguous()
        self.x1 = torch.empty_like(x1, dtype=torch.float32).contiguous()
        self.t = torch.empty_like(t, dtype=torch.float32).contiguous()
        self.graph = torch.cuda.CUDAGraph()
        self.loss = None
