"""hipGraph-captured optimisation steps: ``CFMTrainer(graph=True)`` / ``Pix2PixTrainer(graph=True)``
replay ONE captured graph per step -- forward, loss, backward with the weight gradients forked onto the side stream,
Adam (its scalars read from device memory), repack -- instead of ~200 Python-issued launches.  Same kernels in the same
order, so the replayed steps must equal the eager trainer's BIT FOR BIT: losses and every parameter after every step,
across a learning-rate change (the reference's ReduceLROnPlateau edits ``optimizer.param_groups[0]["lr"]``,
configs/model/*.yaml:12-16) and a change of batch shape (re-capture).  Reference semantics of the step:
src/models/conditional_flow_matching.py:53-88,112-131."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _batches(n, b, hw, seed):
    g = torch.Generator().manual_seed(seed)
    return [((torch.rand(b, 3, hw, hw, generator=g) * 2 - 1).to(DEV), (torch.rand(b, 3, hw, hw, generator=g) * 2 - 1).to(DEV),
             torch.rand(b, generator=g).to(DEV)) for _ in range(n)]


@pytest.mark.parametrize("prec", ["bf16", "fp32"])
def test_cfm_graph_step_is_bit_equal_to_eager(prec):
    from stain2stain_amd import CFMTrainer, FlowUNet
    runs = {}
    data = _batches(5, 4, 64, 3) + _batches(3, 2, 32, 4)         # the shape changes after five steps: a second capture
    for graph in (False, True):
        torch.manual_seed(1984)
        net = FlowUNet(3, [16, 32, 64], 3, 32, precision=prec).to(DEV).train()
        tr = CFMTrainer(net, lr=1e-3, weight_decay=1e-5, graph=graph)
        losses, params = [], []
        for i, (x0, x1, t) in enumerate(data):
            if i == 3:
                tr.optimizer.param_groups[0]["lr"] = 2.5e-4          # what a scheduler does
            losses.append(tr.step(x0, x1, t).clone())
            params.append(tr.flat_p.clone())
        torch.cuda.synchronize()
        runs[graph] = (torch.stack(losses).cpu(), [p.cpu() for p in params],
                       {k: v.detach().cpu().clone() for k, v in net.named_buffers()})
        if graph:
            assert tr._captured is not None and tr.step_count == len(data)
            tr.close()
            assert tr._captured is None
    assert torch.equal(runs[True][0], runs[False][0]), (runs[True][0], runs[False][0])
    for i, (a, b) in enumerate(zip(runs[True][1], runs[False][1])):
        assert torch.equal(a, b), f"parameters differ after step {i}"
    for k, v in runs[False][2].items():                             # BatchNorm running statistics, num_batches_tracked
        assert torch.equal(runs[True][2][k], v), k
    assert float(runs[True][0][-1]) == float(runs[True][0][-1]) and not torch.equal(runs[True][1][0], runs[True][1][-1])


def test_cfm_graph_step_at_production_width_matches_eager_and_takes_fewer_host_calls():
    """Batch 4 of 128x128 tiles through the production widths: three replayed steps equal the eager ones bit for bit."""
    from stain2stain_amd import CFMTrainer, FlowUNet
    data = _batches(4, 4, 128, 5)
    out = {}
    for graph in (False, True):
        torch.manual_seed(1984)
        net = FlowUNet(3, [64, 128, 256, 512, 1024], 3, 256).to(DEV).train()
        tr = CFMTrainer(net, lr=1e-4, weight_decay=1e-5, graph=graph)
        ls = [tr.step(*d).clone() for d in data]
        torch.cuda.synchronize()
        out[graph] = (torch.stack(ls).cpu(), tr.flat_p.cpu())
        tr.close()
        del tr, net
        torch.cuda.empty_cache()
    assert torch.equal(out[True][0], out[False][0]) and torch.equal(out[True][1], out[False][1])


@pytest.mark.parametrize("prec", ["bf16", "fp32"])
def test_pix2pix_graph_step_is_bit_equal_to_eager(prec):
    from stain2stain_amd import PatchGANDiscriminator, Pix2PixGenerator, Pix2PixTrainer
    data = _batches(5, 2, 64, 6)
    runs = {}
    for graph in (False, True):
        torch.manual_seed(1984)
        G, D = Pix2PixGenerator(ngf=16, num_downs=5).to(DEV), PatchGANDiscriminator(ndf=16).to(DEV)
        tr = Pix2PixTrainer(G, D, precision=prec, graph=graph)
        ls, pg, pd = [], [], []
        for i, (src, tgt, _) in enumerate(data):
            if i == 3:
                tr.lr = 5e-5
            ls.append(tr.step(src, tgt).clone())
            pg.append(tr.pG.p.clone()); pd.append(tr.pD.p.clone())
        torch.cuda.synchronize()
        runs[graph] = (torch.stack(ls).cpu(), [p.cpu() for p in pg], [p.cpu() for p in pd])
        if graph:
            assert tr._captured is not None and tr.pG.step_count == tr.pD.step_count == len(data)
            tr.close()
    assert torch.equal(runs[True][0], runs[False][0])
    for i in range(len(data)):
        assert torch.equal(runs[True][1][i], runs[False][1][i]), f"generator differs after step {i}"
        assert torch.equal(runs[True][2][i], runs[False][2][i]), f"discriminator differs after step {i}"


@pytest.mark.parametrize("prec,side", [("bf16", True), ("fp32", True), ("bf16", False)])
def test_pix2pix_update_in_backward_is_bit_equal_to_the_end_of_step_update(prec, side):
    """The generator's Adam slices + repacks issued layer by layer inside the backward pass (on the side stream behind
    each weight gradient; ``Pix2PixTrainer.opt_in_backward``) against one Adam launch + one repack after the join:
    losses, parameters, both Adam moments and the packed MFMA operands after every step."""
    from stain2stain_amd import PatchGANDiscriminator, Pix2PixGenerator, Pix2PixTrainer
    data = _batches(4, 2, 64, 8)
    runs = {}
    for in_bwd in (False, True):
        torch.manual_seed(1984)
        G, D = Pix2PixGenerator(ngf=16, num_downs=5).to(DEV), PatchGANDiscriminator(ndf=16).to(DEV)
        tr = Pix2PixTrainer(G, D, precision=prec)
        tr.opt_in_backward, tr.overlap_wgrad = in_bwd, side
        rec = []
        for src, tgt, _ in data:
            ls = tr.step(src, tgt).clone()
            rec.append((ls, tr.pG.p.clone(), tr.pG.m.clone(), tr.pG.v.clone(), tr.pD.p.clone(),
                        [l.wf.clone() for l in tr.g_down + tr.g_up], [l.wd.clone() for l in tr.g_down + tr.g_up]))
        torch.cuda.synchronize()
        assert tr.pG.step_count == len(data)
        runs[in_bwd] = rec
    for i, (a, b) in enumerate(zip(runs[True], runs[False])):
        for x, y in zip(a[:5], b[:5]):
            assert torch.equal(x, y), f"step {i}"
        for x, y in zip(a[5] + a[6], b[5] + b[6]):
            assert torch.equal(x, y), f"packed operand differs after step {i}"


def test_pix2pix_split_k_folded_in_the_norm_is_bit_equal_to_separate_reduce_launches():
    """The inner levels' split-K convolutions hand their fp32 partial slabs to the InstanceNorm launch that consumes them
    (forward: the encoder / decoder norms; backward: the incoming gradient of the encoder norms) instead of a reduce launch
    of their own: same order of additions, same rounding -- losses, every parameter and the saved activations' effect on
    the next steps bit-equal.  Headline widths at 64 x 64 tiles so that the inner levels really are split."""
    from stain2stain_amd import PatchGANDiscriminator, Pix2PixGenerator, Pix2PixTrainer, pix2pix_engine, ops
    data = _batches(3, 4, 64, 9)
    runs, folded = {}, {}
    for fold in (False, True):
        pix2pix_engine.FOLD_IN_NORM = fold
        count = {"n": 0}
        orig = ops.instnorm_lrelu_fwd2_split

        def counting(*a, _o=orig, **k):
            count["n"] += 1
            return _o(*a, **k)
        ops.instnorm_lrelu_fwd2_split = counting
        try:
            torch.manual_seed(1984)
            G, D = Pix2PixGenerator(ngf=64, num_downs=6).to(DEV), PatchGANDiscriminator(ndf=16).to(DEV)
            tr = Pix2PixTrainer(G, D)
            rec = []
            for src, tgt, _ in data:
                rec.append((tr.step(src, tgt).clone(), tr.pG.p.clone(), tr.pD.p.clone()))
            torch.cuda.synchronize()
        finally:
            ops.instnorm_lrelu_fwd2_split = orig
            pix2pix_engine.FOLD_IN_NORM = True
        runs[fold], folded[fold] = rec, count["n"]
    assert folded[False] == 0 and folded[True] >= 3 * 2, folded            # at least two folded norms per forward pass
    for i, (a, b) in enumerate(zip(runs[True], runs[False])):
        for x, y in zip(a, b):
            assert torch.equal(x, y), f"step {i}"


def test_captured_step_owns_its_workspaces():
    """a captured step bakes the addresses of its split-K / weight-gradient slabs into the graph, and a replay is
    not ordered against a later host-side replacement of a process-global buffer.  A capture therefore allocates its
    workspaces itself (graph-private pool): growing the shared buffers afterwards -- a second, wider trainer on the same
    streams -- and recycling the memory they used to occupy must not change what the replays compute."""
    from stain2stain_amd import CFMTrainer, FlowUNet, ops
    data = _batches(6, 4, 64, 11)

    def run(disturb):
        torch.manual_seed(7)
        net = FlowUNet(3, [16, 32, 64], 3, 32).to(DEV).train()
        tr = CFMTrainer(net, lr=1e-3, weight_decay=1e-5, graph=True)
        out = [tr.step(*data[0]).clone(), tr.step(*data[1]).clone()]          # eager warm-up, then the capture
        assert tr._captured is not None
        if disturb:
            torch.manual_seed(8)
            wide = CFMTrainer(FlowUNet(3, [32, 64, 128, 256], 3, 32).to(DEV).train(), lr=1e-3)
            g = torch.Generator().manual_seed(12)
            big = ((torch.rand(8, 3, 128, 128, generator=g) * 2 - 1).to(DEV), (torch.rand(8, 3, 128, 128, generator=g) * 2 - 1).to(DEV),
                   torch.rand(8, generator=g).to(DEV))
            wide.step(*big)                                                    # grows every shared workspace
            torch.cuda.synchronize()
            del wide
            torch.cuda.empty_cache()
            junk = [torch.full((1 << 22,), float("nan"), device=DEV) for _ in range(16)]      # recycle what was freed
        for d in data[2:]:
            out.append(tr.step(*d).clone())
        torch.cuda.synchronize()
        res = (torch.stack(out).cpu(), tr.flat_p.cpu())
        tr.close()
        return res

    a, b = run(False), run(True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
