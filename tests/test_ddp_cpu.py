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
    # a cap cuts the big layers into pieces that become ready together (round-1 VERDICT: one 55 MB bucket carried 43 %
    # of the bytes): pieces tile the bucket, respect the alignment, none exceeds the cap by more than the alignment
    sizes = [64, 6400, 128, 64000, 640, 64]
    c = plan_buckets(sizes, 256, 10000, 64)
    assert c[0][1] == 0 and c[-1][2] == sum(sizes)
    assert all(hi0 == lo1 and g0 <= g1 for (g0, _, hi0), (g1, lo1, _) in zip(c, c[1:]))
    assert all(hi - lo <= 10000 + 64 and lo % 64 == 0 for _, lo, hi in c)
    assert sum(1 for gi, _, _ in c if gi == 3) == 7            # the 64000-element layer: seven pieces, ready with group 3


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
    closing = []
    for g in range(len(sizes)):
        closing.append(bk.closes(g))
        bk.mark_ready_ordered(g, None)          # (no side stream on CPU: the plain mark_ready)
    ok = ok and closing == [False, True, False, False, True] and not bk.closes(len(sizes) - 1)
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
    # reduce-scatter mode (sharded optimiser): after the exchange a rank owns the summed gradient on its slices, updates
    # the parameters there, and the all-gather leaves every rank with the same full parameter vector
    sizes = [64, 640, 128, 1920, 64]
    n = sum(sizes)
    g = torch.arange(n, dtype=torch.float32) * (rank + 1)
    prm = torch.full((n,), -7.0)
    rs = GradBucketer(g, sizes, bucket_mb=256 * 4 / (1 << 20), max_bucket_mb=1024 * 4 / (1 << 20), mode="reduce_scatter")
    ok = ok and rs.mode == "reduce_scatter" and all((hi - lo) % (8 * world) == 0 for _, lo, hi in rs.buckets)
    rs.start_step()
    for gi in range(len(sizes)):
        rs.mark_ready(gi)
    rs.wait_all()
    total = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
    own = rs.shards()
    ok = ok and len(own) == len(rs.buckets) and sum(hi - lo for lo, hi in own) * world == n
    for lo, hi in own:
        ok = ok and torch.equal(g[lo:hi], total[lo:hi])
        prm[lo:hi] = -g[lo:hi] * rs.grad_scale                 # the rank's "optimiser step" on its slices only
    rs.all_gather(prm)
    ok = ok and torch.equal(prm, -total / world)
    try:
        GradBucketer(torch.zeros(24), [8, 16], mode="reduce_scatter")       # groups must divide into aligned shards
        ok = False
    except ValueError:
        pass
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
    """Backward-completion order of the production net, one group per conv + BatchNorm layer: the big gradients close
    their buckets mid-backward, no bucket exceeds the 16 MB cap (round 1 had one of 55 MB) and what has to wait for the
    very last layer is the stem's 0.15 MB."""
    from stain2stain_amd import FlowUNet
    from stain2stain_amd.ddp import ALIGN, plan_buckets
    from stain2stain_amd.trainer import _param_groups
    net = FlowUNet()
    groups = _param_groups(net)
    assert len(groups) == 20                    # head, 4 x 2 decoder layers, time MLP, 5 x 2 encoder layers
    sizes = [(sum((p.numel() + 7) // 8 * 8 for _, _, p in g) + ALIGN - 1) // ALIGN * ALIGN for g in groups]
    assert sum(p.numel() for g in groups for _, _, p in g) == sum(p.numel() for p in net.parameters()) == 31_785_603
    buckets = plan_buckets(sizes, int(4.0 * (1 << 20) / 4), int(16.0 * (1 << 20) / 4), ALIGN)
    mb = [round((hi - lo) * 4 / 2 ** 20, 2) for _, lo, hi in buckets]
    assert len(buckets) == 14 and mb[-1] < 0.2 and 13 < max(mb) <= 16.0
    assert buckets[-1][0] == len(sizes) - 1 and buckets[0][1] == 0 and buckets[-1][2] == sum(sizes)
    # backward order inside a block: the second conv of a DoubleConv finishes before the first
    names = [[n for _, n, _ in g] for g in groups]
    assert names[1][0].startswith("ups.3.conv.double_conv.3") and names[2][0].startswith("ups.3.conv.double_conv.0")
    assert names[-1][0].startswith("inc.double_conv.0")
