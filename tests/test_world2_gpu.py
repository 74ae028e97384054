"""World size 2 on the hardware: two ranks (two processes sharing the one GPU of the test box, ``gloo`` transport -- RCCL
refuses two ranks on one device) run the data-parallel training steps through the HIP kernels, and the parent checks
them against single-process runs of the same global batch.

What this pins, that the CPU gloo tests of the host logic (tests/test_ddp_cpu.py) and the forced world-size-1 RCCL test
(tests/test_bench_contract_gpu.py) cannot:
  * replicas initialised differently are equal after the constructor's broadcast and stay BIT-equal over optimiser steps,
  * the bucketed all-reduce + 1/world in Adam equals the mean of the per-rank gradients (per-replica BatchNorm, the
    stain experiments' setting: configs/experiment/gray_matter/simple_flow_matching.yaml:18-23),
  * reduce-scatter + sharded Adam + parameter all-gather gives bit-identical parameters,
  * SyncBatchNorm (configs/trainer/ddp.yaml:9): two ranks x 4 tiles == one process x 8 tiles (full-batch statistics),
    by the trainer's flag and by torch.nn.SyncBatchNorm.convert_sync_batchnorm on the module (Lightning's route),
  * the pix2pix G + D step (InstanceNorm: per sample): two ranks x 2 tiles == one process x 4 tiles.
Reference behaviour: DDP mean of gradients (configs/trainer/ddp.yaml:4), rank r takes ``batch_size // world_size`` tiles
(src/data/paired_data_module.py:273-278)."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu

CH, TILE, PER_RANK = [16, 32], 64, 4
LR = 1e-4


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _data(n, seed=1984):
    g = torch.Generator().manual_seed(seed)
    x0 = torch.rand(n, 3, TILE, TILE, generator=g) * 2 - 1
    x1 = torch.rand(n, 3, TILE, TILE, generator=g) * 2 - 1
    t = torch.rand(2, n, generator=g)
    return x0, x1, t


def _unet(seed):
    from stain2stain_amd import FlowUNet
    torch.manual_seed(seed)
    return FlowUNet(3, CH, 3, 32, precision="fp32").to("cuda:0").train()


def _p2p_nets(seed):
    from stain2stain_amd import PatchGANDiscriminator, Pix2PixGenerator
    torch.manual_seed(seed)
    return Pix2PixGenerator(ngf=16, num_downs=5).to("cuda:0"), PatchGANDiscriminator(ndf=16).to("cuda:0")


def _buffers(net):
    return {k: v.detach().cpu().clone() for k, v in net.named_buffers()}


def _cfm_steps(trainer, net, x0, x1, t, steps=2):
    out = {"loss": [], "grad": None}
    for s in range(steps):
        loss, _ = trainer.forward_backward(x0.cuda(), x1.cuda(), t[s].cuda(), want_v=False)
        if s == 0:
            trainer.bucketer.wait_all()
            torch.cuda.synchronize()
            out["grad"] = (trainer.flat_g * trainer.bucketer.grad_scale).cpu()     # (optimizer_step's own wait_all then
                                                                                   # finds nothing left to join)
        trainer.optimizer_step()
        out["loss"].append(float(loss))
    torch.cuda.synchronize()
    out["param"] = trainer.flat_p.cpu()
    out["buffers"] = _buffers(net)
    return out


def _rank_main(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from stain2stain_amd import CFMTrainer, Pix2PixTrainer
    res = {}
    x0, x1, t = _data(world * PER_RANK)
    lo, hi = rank * PER_RANK, (rank + 1) * PER_RANK
    for name, kw in (("ddp", {}), ("sharded", {"sharded_optimizer": True}), ("syncbn", {"sync_batchnorm": True}),
                     ("converted", {})):
        net = _unet(100 + rank)                    # replicas start DIFFERENT: the constructor broadcasts rank 0's
        if name == "converted":                    # what Lightning does for `sync_batchnorm: True`
            net = torch.nn.SyncBatchNorm.convert_sync_batchnorm(net)
        tr = CFMTrainer(net, lr=LR, weight_decay=1e-5, **kw)
        assert tr.bucketer.enabled and tr.bucketer.world == world
        res[name] = _cfm_steps(tr, net, x0[lo:hi], x1[lo:hi], t[:, lo:hi])
        if name in ("ddp", "sharded"):
            # checkpoint -> fresh trainer -> one more step (a sharded rank's Adam moments are only current on
            # its own shards; the dump must gather them) against the original trainer simply continuing
            t3 = torch.rand(world * PER_RANK, generator=torch.Generator().manual_seed(7))[lo:hi]
            sd = tr.optimizer_state_dict()              # every rank calls: a collective in sharded mode
            weights = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
            net2 = _unet(300 + rank)
            tr2 = CFMTrainer(net2, lr=LR, weight_decay=1e-5, **kw)
            net2.load_state_dict(weights)
            tr2.load_optimizer_state_dict(sd)
            for trx, key in ((tr, "continued"), (tr2, "resumed")):
                trx.forward_backward(x0[lo:hi].cuda(), x1[lo:hi].cuda(), t3.cuda(), want_v=False)
                trx.optimizer_step()
                torch.cuda.synchronize()
                res[name][key] = trx.flat_p.cpu()
            res[name]["moments_nonzero"] = bool(all(float(s["exp_avg_sq"].abs().max()) > 0 for s in sd["state"].values()))
    G, D = _p2p_nets(200 + rank)
    p2p = Pix2PixTrainer(G, D, precision="fp32", sync_loss=False)
    src, tgt = x0[:4][2 * rank: 2 * rank + 2], x1[:4][2 * rank: 2 * rank + 2]
    losses = p2p.step(src.cuda(), tgt.cuda())
    torch.cuda.synchronize()
    res["p2p"] = {"pG": p2p.pG.p.cpu(), "pD": p2p.pD.p.cpu(), "losses": losses.cpu()}
    # the same for the pix2pix trainer with the sharded optimiser: step, dump, load into a fresh trainer, step both
    G, D = _p2p_nets(200 + rank)
    sh = Pix2PixTrainer(G, D, precision="fp32", sync_loss=False, sharded_optimizer=True)
    sh.step(src.cuda(), tgt.cuda())
    torch.cuda.synchronize()
    res["p2p"]["sharded_pG"] = sh.pG.p.cpu()
    sd = sh.optimizer_state_dict()
    G2, D2 = _p2p_nets(400 + rank)
    sh2 = Pix2PixTrainer(G2, D2, precision="fp32", sync_loss=False, sharded_optimizer=True)
    G2.load_state_dict({k: v.detach().cpu().clone() for k, v in G.state_dict().items()})
    D2.load_state_dict({k: v.detach().cpu().clone() for k, v in D.state_dict().items()})
    sh2.packG.repack(); sh2.packD.repack()
    sh2.load_optimizer_state_dict(sd)
    for trx, key in ((sh, "continued"), (sh2, "resumed")):
        trx.step(tgt.cuda(), src.cuda())
        torch.cuda.synchronize()
        res["p2p"][key] = (trx.pG.p.cpu(), trx.pD.p.cpu())
    torch.save(res, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.fixture(scope="module")
def ranks(tmp_path_factory):
    import torch.multiprocessing as mp
    out_dir = str(tmp_path_factory.mktemp("world2"))
    mp.spawn(_rank_main, args=(2, _free_port(), out_dir), nprocs=2, join=True)
    return [torch.load(os.path.join(out_dir, f"rank{r}.pt"), weights_only=True) for r in range(2)]


def _close(a, b, rel, what):
    err = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)
    assert err <= rel, (what, err)


def test_replicas_stay_bit_equal(ranks):
    for name in ("ddp", "sharded", "syncbn"):
        assert torch.equal(ranks[0][name]["param"], ranks[1][name]["param"]), name
        assert torch.equal(ranks[0][name]["grad"], ranks[1][name]["grad"]) or name == "sharded"
    assert torch.equal(ranks[0]["p2p"]["pG"], ranks[1]["p2p"]["pG"]) and torch.equal(ranks[0]["p2p"]["pD"], ranks[1]["p2p"]["pD"])
    # reduce-scatter + Adam on 1/world of every bucket + all-gather: the same parameters as all-reduce + full Adam
    assert torch.equal(ranks[0]["sharded"]["param"], ranks[0]["ddp"]["param"])
    assert ranks[0]["sharded"]["loss"] == ranks[0]["ddp"]["loss"]


def test_checkpointed_optimizer_state_resumes_bit_equal(ranks):
    """Dump -> fresh trainer -> load -> one more step equals the original trainer continuing, in both exchange modes
    (the sharded dump gathers the moments of the shards the other rank owns), and both modes agree with each other."""
    for r in (0, 1):
        for name in ("ddp", "sharded"):
            assert ranks[r][name]["moments_nonzero"], name
            assert torch.equal(ranks[r][name]["resumed"], ranks[r][name]["continued"]), (r, name)
        assert torch.equal(ranks[r]["sharded"]["continued"], ranks[r]["ddp"]["continued"])
        for k in (0, 1):
            assert torch.equal(ranks[r]["p2p"]["resumed"][k], ranks[r]["p2p"]["continued"][k]), (r, k)
    assert torch.equal(ranks[0]["sharded"]["resumed"], ranks[1]["sharded"]["resumed"])
    assert torch.equal(ranks[0]["p2p"]["sharded_pG"], ranks[0]["p2p"]["pG"])       # sharded == all-reduce, pix2pix too
    assert torch.equal(ranks[0]["p2p"]["resumed"][0], ranks[1]["p2p"]["resumed"][0])


def test_allreduce_is_the_mean_of_the_rank_gradients(ranks):
    """Per-replica BatchNorm (the stain experiments): one process evaluates the two half batches one after the other from
    rank 0's initialisation, averages the gradients and steps -- what the two ranks did between them."""
    from stain2stain_amd import CFMTrainer
    x0, x1, t = _data(2 * PER_RANK)
    net = _unet(100)
    tr = CFMTrainer(net, lr=LR, weight_decay=1e-5)
    assert not tr.bucketer.enabled
    bufs = list(net.buffers())
    for s in range(2):
        gs, losses = [], []
        for r in (1, 0):                                  # rank 0's half last: its running statistics are the ones kept
            keep = [b.clone() for b in bufs]
            loss, _ = tr.forward_backward(x0[r * 4:r * 4 + 4].cuda(), x1[r * 4:r * 4 + 4].cuda(), t[s, r * 4:r * 4 + 4].cuda(),
                                          want_v=False)
            gs.append(tr.flat_g.clone()); losses.append(float(loss))
            if r == 1:
                for b, k in zip(bufs, keep):
                    b.copy_(k)
        tr.flat_g.copy_((gs[0] + gs[1]) * 0.5)
        if s == 0:
            _close(ranks[0]["ddp"]["grad"], tr.flat_g.cpu(), 2e-5, "step-1 gradient")
        tr.optimizer_step()
        for r in (0, 1):
            assert abs(ranks[r]["ddp"]["loss"][s] - losses[1 - r]) <= (1e-5 if s == 0 else 1e-3) * abs(losses[1 - r]), (s, r)
    torch.cuda.synchronize()
    d = (ranks[0]["ddp"]["param"] - tr.flat_p.cpu()).abs()
    # Adam's first steps move every weight by ~lr whatever the gradient's size: a gradient that is zero to rounding may
    # take either sign, so single elements can differ by 2 lr per step; everything else agrees to rounding
    assert float(d.max()) <= 4.5 * LR and float(d.mean()) <= 2e-7, (float(d.max()), float(d.mean()))
    for k, v in _buffers(net).items():
        _close(ranks[0]["ddp"]["buffers"][k].float(), v.float(), 1e-5, k)


def test_syncbatchnorm_two_ranks_equal_one_process_on_the_global_batch(ranks):
    from stain2stain_amd import CFMTrainer
    x0, x1, t = _data(2 * PER_RANK)
    net = _unet(100)
    tr = CFMTrainer(net, lr=LR, weight_decay=1e-5)
    ref = _cfm_steps(tr, net, x0, x1, t)
    _close(ranks[0]["syncbn"]["grad"], ref["grad"], 2e-5, "step-1 gradient")
    for s in range(2):
        mean_loss = 0.5 * (ranks[0]["syncbn"]["loss"][s] + ranks[1]["syncbn"]["loss"][s])
        assert abs(mean_loss - ref["loss"][s]) <= (1e-5 if s == 0 else 1e-3) * abs(ref["loss"][s]), s
    d = (ranks[0]["syncbn"]["param"] - ref["param"]).abs()
    assert float(d.max()) <= 4.5 * LR and float(d.mean()) <= 2e-7, (float(d.max()), float(d.mean()))
    for r in (0, 1):                      # running statistics of the GLOBAL batch on every rank
        for k, v in ref["buffers"].items():
            _close(ranks[r]["syncbn"]["buffers"][k].float(), v.float(), 1e-5, k)
    # Lightning's route -- torch.nn.SyncBatchNorm.convert_sync_batchnorm on the module, no trainer flag -- is the same step
    for r in (0, 1):
        assert torch.equal(ranks[r]["converted"]["param"], ranks[r]["syncbn"]["param"])
        assert ranks[r]["converted"]["loss"] == ranks[r]["syncbn"]["loss"]
    # and it is not what per-replica statistics give
    assert float((ranks[0]["syncbn"]["grad"] - ranks[0]["ddp"]["grad"]).abs().max()) > 1e-3 * float(ref["grad"].abs().max())


def test_pix2pix_two_ranks_equal_one_process_on_the_global_batch(ranks):
    from stain2stain_amd import Pix2PixTrainer
    x0, x1, _ = _data(2 * PER_RANK)
    G, D = _p2p_nets(200)
    p2p = Pix2PixTrainer(G, D, precision="fp32", sync_loss=False)
    losses = p2p.step(x0[:4].cuda(), x1[:4].cuda())
    torch.cuda.synchronize()
    both = 0.5 * (ranks[0]["p2p"]["losses"] + ranks[1]["p2p"]["losses"])
    _close(both, losses.cpu(), 2e-5, "losses")
    for k, ref in (("pG", p2p.pG.p.cpu()), ("pD", p2p.pD.p.cpu())):
        d = (ranks[0]["p2p"][k] - ref).abs()
        assert float(d.max()) <= 2.2 * 2e-4 and float(d.mean()) <= 1e-6, (k, float(d.max()), float(d.mean()))
