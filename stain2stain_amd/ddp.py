"""Data-parallel gradient exchange: one flat fp32 gradient buffer, bucketed all-reduce over RCCL.

Replaces what Lightning's DDPStrategy + torch DDP do for the reference (``strategy: ddp``,
configs/trainer/ddp.yaml:4): every rank holds a full replica and the only
data-path collective is the gradient all-reduce (mean).  The flat buffer is laid out in the order in
which the backward pass completes gradients (decoder head first, encoder stem last), so a bucket is a
contiguous slice that becomes ready while earlier layers are still being differentiated;
``mark_ready`` launches the all-reduce of every newly complete bucket asynchronously (on RCCL's own
stream, ordered after the compute stream's work so far) and ``wait_all`` joins them before the
optimiser step.  The averaging factor 1/world_size is folded into the Adam kernel's ``grad_scale``.

Pure host logic over ``torch.distributed``: works with the ``nccl`` (= RCCL) backend on GPUs and with
``gloo`` on CPU tensors, which is how tests/test_ddp_cpu.py exercises world_size = 2.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def plan_buckets(group_sizes: Sequence[int], bucket_elems: int) -> List[Tuple[int, int, int]]:
    """Merge consecutive parameter groups into buckets of at least ``bucket_elems`` elements.

    Returns [(first_group, start_offset, end_offset)] with buckets covering [0, sum(group_sizes)); a
    bucket is ready once all groups up to its last one are done, recorded as ``last_group`` in the
    first slot of the following bucket minus one (see GradBucketer).
    """
    buckets = []
    start, acc, first = 0, 0, 0
    off = 0
    for gi, n in enumerate(group_sizes):
        off += n
        acc += n
        if acc >= bucket_elems or gi == len(group_sizes) - 1:
            buckets.append((gi, start, off))   # ready after group gi
            start, acc = off, 0
    return buckets


class GradBucketer:
    def __init__(self, flat_grad: torch.Tensor, group_sizes: Sequence[int], bucket_mb: float = 32.0,
                 process_group: Optional[dist.ProcessGroup] = None):
        self.flat = flat_grad
        self.pg = process_group
        # S2S_FORCE_DDP=1 issues the collectives even at world size 1 (exercises the RCCL path on a one-GPU box)
        self.enabled = dist.is_available() and dist.is_initialized() and (
            dist.get_world_size(process_group) > 1 or os.environ.get("S2S_FORCE_DDP") == "1")
        self.world = dist.get_world_size(process_group) if self.enabled else 1
        if sum(group_sizes) != flat_grad.numel():
            raise ValueError("group sizes do not cover the flat gradient buffer")
        self.buckets = plan_buckets(group_sizes, int(bucket_mb * (1 << 20) / flat_grad.element_size()))
        self._next = 0
        self._works: List = []

    @property
    def grad_scale(self) -> float:
        """Factor that turns the all-reduced SUM into DDP's mean."""
        return 1.0 / self.world

    def start_step(self) -> None:
        self._next = 0
        self._works = []

    def mark_ready(self, group_index: int) -> None:
        """Groups 0..group_index have their gradients written (in stream order): reduce complete buckets."""
        while self._next < len(self.buckets) and self.buckets[self._next][0] <= group_index:
            _, lo, hi = self.buckets[self._next]
            if self.enabled:
                self._works.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.pg,
                                                   async_op=True))
            self._next += 1

    def wait_all(self) -> None:
        if self._next != len(self.buckets):
            raise RuntimeError("wait_all() before every gradient group was marked ready")
        for w in self._works:
            w.wait()
        self._works = []


def _active(process_group) -> bool:
    return dist.is_available() and dist.is_initialized() and (
        dist.get_world_size(process_group) > 1 or os.environ.get("S2S_FORCE_DDP") == "1")


def broadcast_from_rank0(tensors: Sequence[torch.Tensor], process_group=None) -> None:
    """Initial replica synchronisation (what torch DDP does at construction)."""
    if not _active(process_group):
        return
    for t in tensors:
        dist.broadcast(t, src=0, group=process_group)


def all_reduce_mean_scalar(x: torch.Tensor, process_group=None):
    """``self.log(..., sync_dist=True)`` of the reference (conditional_flow_matching.py:86): mean over ranks."""
    if not _active(process_group):
        return None
    return dist.all_reduce(x, op=dist.ReduceOp.SUM, group=process_group, async_op=True)
