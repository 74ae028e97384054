"""The driver's contract for bench.py: one JSON line on stdout with the agreed keys (short run, GPU)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra, env=None, steps=2):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", "1", *extra],
                         capture_output=True, text=True, timeout=600, cwd=ROOT,
                         env=None if env is None else {**os.environ, **env})
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_training_line_has_the_contract_keys():
    d = _run()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    # BASELINE.json's metric is worded on the pix2pix G + D step: the driver's line carries it under its own key
    p = d["pix2pix"]
    assert "pix2pix" in p["metric"] and p["unit"] == "tiles/s" and p["value"] > 100 and p["roofline"]["bound"] == "mfma"
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "bf16" and d["unit"] == "tiles/s"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["achieved"] > 100
    assert abs(d["value"] - 16 * 1000.0 / d["ms_per_step"]) < 0.01 * d["value"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    # eager launches in the headline loop, the hipGraph replay of the same trainer timed beside it
    assert d["config"]["launch"].startswith("eager") and d["config"]["graph_ms_per_step"] > 0
    assert p["config"]["launch"].startswith("eager") and p["config"]["graph_ms_per_step"] > 0
    # the driver's default line times what DESIGN quotes -- configs[3] (batch 32 x 50 Euler steps, eager
    # and graphed, and the reference's batch 1) and the fp32 parity mode of the headline step
    sm = d["sample"]
    for b in ("batch32", "batch1"):
        for how in ("eager", "graph"):
            r = sm[b][how]
            assert r["finite"] is True and r["tiles_per_s"] > 1 and abs(r["ms_per_euler_step"] * 50 - r["ms_per_solve"]) < 0.05
    assert sm["value"] == sm["batch32"]["graph"]["tiles_per_s"] and sm["batch1"]["graph"]["ms_per_euler_step"] < 2.0
    fp = d["fp32_parity"]
    assert fp["unit"] == "tiles/s" and 50 < fp["value"] < d["value"] and fp["dtype"].startswith("fp32")


def test_dropin_path_is_within_ten_percent_of_the_fused_trainer():
    """what a user of the reference gets after the ``_target_`` swap -- FlowUNet under autograd inside
    ConditionalFlowMatchingModule, zero_grad / training_step / backward / optimizer.step with stain2stain_amd.FusedAdam --
    timed in the default line under the headline's contract, at most 1.10 x the fused trainer's step."""
    d = _run("--no-pix2pix", "--no-cpu-baseline", steps=16)
    dr = d["dropin"]
    assert dr["unit"] == "tiles/s" and dr["dtype"] == "bf16" and "training_step" in dr["metric"]
    assert dr["fused_adam"]["ms_per_step"] > 0 and dr["torch_adam"]["ms_per_step"] > 0
    assert abs(dr["vs_fused_trainer"] - dr["ms_per_step"] / d["ms_per_step"]) < 1e-3
    assert dr["vs_fused_trainer"] <= 1.10, dr
    assert 0 < dr["fused_adam"]["final_loss"] < 2 and 0 < dr["torch_adam"]["final_loss"] < 2


def test_sampling_line():
    d = _run("--mode", "sample", "--euler-steps", "3", "--no-cpu-baseline")
    assert d["unit"] == "tiles/s" and d["config"]["finite"] is True and d["roofline"]["achieved"] > 100


def test_sampling_line_full_50_euler_steps_batch_32():
    """BASELINE.json configs[3] as worded: batch 32, 50 Euler steps, once eagerly and once as replayed graphs."""
    d = _run("--mode", "sample", "--euler-steps", "50", "--no-cpu-baseline", steps=1)
    assert d["config"]["finite"] is True and d["config"]["global_batch"] == 32 and d["value"] > 50
    assert abs(d["config"]["ms_per_euler_step"] * 50 - d["ms_per_step"]) < 0.05 * d["ms_per_step"]
    g = _run("--mode", "sample", "--euler-steps", "50", "--no-cpu-baseline", "--graph", steps=1)
    assert g["config"]["finite"] is True and g["value"] > 50


def test_pix2pix_line():
    """Separate line for the pix2pix G + D step (row a13 is not in the reference)."""
    d = _run("--mode", "pix2pix")
    assert d["unit"] == "tiles/s" and "pix2pix" in d["metric"] and d["n_gpus"] == 1 and d["vs_baseline"] is None
    assert d["roofline"]["bound"] == "mfma" and d["roofline"]["achieved"] > 10
    assert d["config"]["loss_d"] == d["config"]["loss_d"] and d["config"]["loss_g"] == d["config"]["loss_g"]   # finite


def test_rccl_path_in_a_fresh_process_matches_the_single_process_run():
    """the gradient exchange has to run over RCCL somewhere in the GPU suite.  A fresh child process
    runs bench.py with S2S_FORCE_DDP=1 (process group "nccl" = RCCL at world size 1: bucketed all-reduce on RCCL's
    stream, joined before Adam, 1/world folded into the update) and its final losses -- CFM step and pix2pix G + D step --
    must equal the run without any collective BIT FOR BIT; the reduce-scatter / sharded-Adam / all-gather mode as well."""
    port = {"MASTER_ADDR": "127.0.0.1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    plain = _run("--no-cpu-baseline", steps=3)
    ddp = _run("--no-cpu-baseline", steps=3, env={"S2S_FORCE_DDP": "1", "MASTER_PORT": "29611", **port})
    rs = _run("--no-cpu-baseline", "--sharded-optimizer", steps=3, env={"S2S_FORCE_DDP": "1", "MASTER_PORT": "29612", **port})
    assert plain["config"]["grad_exchange"] == "none" and ddp["config"]["grad_exchange"] == "allreduce"
    assert rs["config"]["grad_exchange"] == "reduce_scatter"
    assert max(ddp["config"]["buckets_mb"]) <= 16.0 and len(ddp["config"]["buckets_mb"]) == 14
    for other in (ddp, rs):
        assert other["config"]["loss_bits"] == plain["config"]["loss_bits"]
        assert other["pix2pix"]["config"]["loss_bits"] == plain["pix2pix"]["config"]["loss_bits"]
    assert ddp["pix2pix"]["config"]["grad_exchange"] == "allreduce" and max(ddp["pix2pix"]["config"]["buckets_mb"]) <= 16.0
    # SyncBatchNorm's small all-reduces on RCCL (two per layer and step): at world size 1 they are identities, the
    # statistics go through the exchange form (float sums, one-block finalize), so the loss agrees to rounding
    syn = _run("--no-cpu-baseline", "--no-pix2pix", "--sync-batchnorm", steps=3,
               env={"S2S_FORCE_DDP": "1", "MASTER_PORT": "29613", **port})
    assert syn["config"]["sync_batchnorm"] is True and plain["config"]["sync_batchnorm"] is False
    assert abs(syn["config"]["final_loss"] - plain["config"]["final_loss"]) <= 2e-3 * plain["config"]["final_loss"]


def test_two_rank_launch_as_the_driver_does_it():
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...` (the driver's multi-GPU command) on the
    one-GPU test box: both ranks share the device and the transport is gloo (S2S_BENCH_BACKEND; RCCL wants a device per
    rank), everything else -- env rendezvous, per-rank shards, barriers, max-over-ranks clock, rank 0's single JSON line,
    the bucketed gradient exchange behind the HIP backward -- is the code the 8-GPU run takes."""
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29631", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-pix2pix"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT,
                         env={**os.environ, "S2S_BENCH_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 32 and d["config"]["parallelism"] == "dp2"
    assert d["scaling"] == "weak" and d["config"]["grad_exchange"] == "allreduce" and "cpu_baseline" not in d
    assert abs(d["value"] - 32 * 1000.0 / d["ms_per_step"]) < 0.01 * d["value"] and d["config"]["final_loss"] > 0
