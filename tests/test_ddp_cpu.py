"""Data-parallel host logic on CPU with gloo, world_size = 2: bucket planning, async all-reduce of the flat
gradient buffer as groups become ready, mean via grad_scale, initial broadcast."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def test_plan_buckets_covers_everything():
    from stain2stain_amd.ddp import plan_buckets
    sizes = [10, 300, 5, 700, 20, 1]
    b = plan_buckets(sizes, 256)
    assert b[0][1] == 0 and b[-1][2] == sum(sizes)
    for (g0, lo0, hi0), (g1, lo1, hi1) in zip(b, b[1:]):
        assert hi0 == lo1 and g0 < g1
    assert b[-1][0] == len(sizes) - 1
    assert plan_buckets([5], 100) == [(0, 0, 5)]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from stain2stain_amd.ddp import GradBucketer, all_reduce_mean_scalar, broadcast_from_rank0
    sizes = [8, 120, 16, 64, 40]
    flat = torch.arange(sum(sizes), dtype=torch.float32) * (rank + 1)
    bk = GradBucketer(flat, sizes, bucket_mb=128 * 4 / (1 << 20))   # 128-element buckets
    ok = bk.enabled and bk.world == world and len(bk.buckets) >= 2
    bk.start_step()
    try:
        bk.wait_all()
        ok = False
    except RuntimeError:
        pass
    for g in range(len(sizes)):
        bk.mark_ready(g)
    bk.wait_all()
    expect = torch.arange(sum(sizes), dtype=torch.float32) * sum(r + 1 for r in range(world))
    ok = ok and torch.equal(flat, expect) and abs(bk.grad_scale - 1.0 / world) < 1e-12
    p = torch.full((7,), float(rank + 5))
    broadcast_from_rank0([p])
    ok = ok and torch.equal(p, torch.full((7,), 5.0))
    l = torch.tensor(float(rank))
    w = all_reduce_mean_scalar(l)
    w.wait()
    ok = ok and float(l) == sum(range(world))
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_bucketed_allreduce_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_single_process_bucketer_is_a_noop():
    from stain2stain_amd.ddp import GradBucketer
    flat = torch.ones(10)
    bk = GradBucketer(flat, [4, 6])
    bk.start_step(); bk.mark_ready(1); bk.wait_all()
    assert not bk.enabled and bk.grad_scale == 1.0 and torch.equal(flat, torch.ones(10))


def test_production_bucket_plan_leaves_only_the_stem_for_the_end():
    """Backward-completion order of the production net: the big gradients (decoder ups.0, encoder downs.3) close their
    buckets mid-backward; what has to wait for the very last layer is the stem's 0.1 MB."""
    from stain2stain_amd import FlowUNet
    from stain2stain_amd.ddp import plan_buckets
    from stain2stain_amd.trainer import _param_groups
    net = FlowUNet()
    groups = _param_groups(net)
    sizes = [sum((p.numel() + 7) // 8 * 8 for _, _, p in g) for g in groups]
    assert sum(p.numel() for g in groups for _, _, p in g) == sum(p.numel() for p in net.parameters()) == 31_785_603
    buckets = plan_buckets(sizes, int(4.0 * (1 << 20) / 4))
    mb = [round((hi - lo) * 4 / 2 ** 20, 1) for _, lo, hi in buckets]
    assert len(buckets) == 6 and mb[-1] < 0.2 and max(mb) > 50
    assert buckets[-1][0] == len(sizes) - 1 and buckets[0][1] == 0 and buckets[-1][2] == sum(sizes)
