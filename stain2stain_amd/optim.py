"""``FusedAdam``: torch.optim.Adam's update (coupled L2 weight decay, no amsgrad -- what the reference builds in
``configure_optimizers``, src/models/conditional_flow_matching.py:112-131 with configs/model/*.yaml:3-7 ``lr: 1e-4,
weight_decay: 1e-5``) over ORDINARY module parameters as one HIP launch per step.

The drop-in path (the reference's LightningModule surface driving ``stain2stain_amd`` modules under autograd) ends every
step in ``optimizer.step()``.  ``torch.optim.Adam`` runs that as a dozen ``_foreach`` passes over 134 tensors; this class
takes the same ``params`` / hyper-parameters, keeps ``exp_avg`` / ``exp_avg_sq`` / ``step`` per parameter under the names
torch uses (so ``state_dict()`` moves between the two and through Lightning checkpoints) and updates every parameter with a
gradient in one launch of ``s2s_adam_multi`` -- element for element the arithmetic of the fused trainer's flat Adam.  A
Hydra config selects it with ``optimizer: {_target_: stain2stain_amd.FusedAdam, _partial_: true, lr: 1e-4,
weight_decay: 1e-5}``.  GPU only: CPU parameters raise.
"""
from __future__ import annotations

from collections import deque
from typing import Dict, List

import torch

from . import _native, engine, ops


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 amsgrad: bool = False, maximize: bool = False):
        if amsgrad or maximize:
            raise ValueError("stain2stain_amd.FusedAdam: amsgrad / maximize are not supported")
        if lr < 0.0 or eps < 0.0 or not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0) or weight_decay < 0.0:
            raise ValueError("stain2stain_amd.FusedAdam: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False,
                                      maximize=False))
        self._desc_cache: Dict[tuple, tuple] = {}
        self._pinned = deque(maxlen=8)        # host descriptor tables whose asynchronous copies may still be in flight

    def _table(self, rows: List[list], dev: torch.device) -> torch.Tensor:
        host = torch.tensor(rows, dtype=torch.int64).pin_memory()
        self._pinned.append(host)
        return host.to(dev, non_blocking=True)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = ops._L()
        for gi, group in enumerate(self.param_groups):
            by_step: Dict[int, list] = {}
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("stain2stain_amd.FusedAdam: parameters must be contiguous fp32 GPU tensors "
                                       "(HIP-only implementation)")
                g = p.grad
                if g.is_sparse:
                    raise RuntimeError("stain2stain_amd.FusedAdam does not support sparse gradients")
                if g.dtype != torch.float32 or not g.is_contiguous():
                    g = g.float().contiguous()
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                by_step.setdefault(int(st["step"]), []).append((p, g, st))
            for step, items in by_step.items():
                dev = items[0][0].device
                # the descriptor table is rebuilt (one small pinned-memory copy) only when a pointer moved: with
                # zero_grad(set_to_none=True) the caching allocator hands the same gradient blocks out step after step
                key = tuple((p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel())
                            for p, g, st in items)
                hit = self._desc_cache.get(gi)
                if hit is None or hit[0] != key:
                    rows, start = [], 0
                    for p, g, st in items:
                        rows.append([p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                     p.numel(), start])
                        start += L.s2s_adam_multi_blocks(p.numel())
                    hit = (key, self._table(rows, dev), start)
                    self._desc_cache[gi] = hit
                with torch.cuda.device(dev):
                    rc = L.s2s_adam_multi(hit[1].data_ptr(), len(items), int(hit[2]), int(step), float(group["lr"]),
                                          float(group["betas"][0]), float(group["betas"][1]), float(group["eps"]),
                                          float(group["weight_decay"]), 1.0, ops._stream())
                _native.check(rc, "adam_multi")
                for p, _, _ in items:          # the kernel wrote through raw pointers: tell torch (packed-weight caches)
                    torch.autograd.graph.increment_version(p)
        engine.mutation_epoch[0] += 1
        return loss
