"""Forward parity at the headline configuration (production widths, batch 16, 256x256 by default) against the CPU
oracle: the fp32 (split-bf16) mode, the bf16 throughput mode, and - as the yardstick for the latter - the oracle
itself with every conv operand and conv output rounded to bf16 (what bf16 storage does to the reference's own
arithmetic, fp32 accumulation).  Prints rel-L2 / max-norm errors of the velocity; B and TILE from the environment."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from stain2stain_amd import CFMTrainer, FlowUNet
from oracle import unet_oracle as O

B, TILE = int(os.environ.get("B", 16)), int(os.environ.get("TILE", 256))
g = torch.Generator().manual_seed(1984)
x0 = torch.rand(B, 3, TILE, TILE, generator=g) * 2 - 1
x1 = torch.rand(B, 3, TILE, TILE, generator=g) * 2 - 1
t = torch.rand(B, generator=g)

NO_HIP = bool(os.environ.get("NO_HIP"))      # CPU-only: the oracle's own fp32-vs-fp64 yardstick (GRADS=2 FP64=1)
res = {}
for prec in ("fp32", "bf16"):
    torch.manual_seed(1984)
    net = FlowUNet(precision=prec)
    if prec == "fp32":
        P = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    if NO_HIP:
        continue
    net = net.cuda().train()
    tr = CFMTrainer(net)
    loss, v = tr.forward_backward(x0.cuda(), x1.cuda(), t.cuda())
    res[prec] = (float(loss), v.detach().float().cpu(),
                 {k: p.grad.detach().float().cpu().clone() for k, p in net.named_parameters()})
    del tr, net

xt, ut = O.cfm_sample(x0, x1, t)
torch.set_num_threads(min(16, os.cpu_count() or 8))     # the GPU box's share is 16 cores whatever it reports
real_conv = F.conv2d
rb = lambda a: a.to(torch.bfloat16).float()
if not NO_HIP and not os.environ.get('SKIP_FWD'):
    with torch.no_grad():
        t0 = time.time()
        v_ref = O.flow_forward(t, xt, P, True)
        print(f"oracle forward {time.time() - t0:.1f} s, loss {float(O.cfm_loss(v_ref, ut)):.6f}", flush=True)
        real_conv = F.conv2d
        rb = lambda a: a.to(torch.bfloat16).float()

        def conv_bf16_storage(x, w, b=None, *args, **kw):
            return rb(real_conv(rb(x), rb(w), b, *args, **kw))
        F.conv2d = conv_bf16_storage
        try:
            v_emu = O.flow_forward(t, xt, P, True)
        finally:
            F.conv2d = real_conv


    def err(a, b):
        return f"rel-L2 {float((a - b).norm() / b.norm()):.3e}  max-norm {float((a - b).abs().max() / b.abs().max()):.3e}"


    print("HIP fp32 mode      vs oracle :", err(res["fp32"][1], v_ref), " loss", res["fp32"][0])
    print("HIP bf16 mode      vs oracle :", err(res["bf16"][1], v_ref), " loss", res["bf16"][0])
    print("oracle bf16-storage vs oracle:", err(v_emu, v_ref))
    print("HIP bf16 mode vs oracle bf16-storage:", err(res["bf16"][1], v_emu))

if os.environ.get("GRADS") == "1":
    # gradients: the oracle's autograd with bf16 storage of activations AND of the gradients flowing back
    class Store(torch.autograd.Function):
        @staticmethod
        def forward(ctx, a):
            return rb(a)

        @staticmethod
        def backward(ctx, g_):
            return rb(g_)

    def conv_bf16_storage_ag(x, w, b=None, *args, **kw):
        return Store.apply(real_conv(Store.apply(x), rb(w.detach()) + (w - w.detach()), b, *args, **kw))
    F.conv2d = conv_bf16_storage_ag
    try:
        t0 = time.time()
        _, _, g_emu, _ = O.loss_and_grads(P, x0, x1, t)
        print(f"oracle fwd+bwd with bf16 storage {time.time() - t0:.1f} s", flush=True)
    finally:
        F.conv2d = real_conv
    cos = lambda a, b: float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
    print(f"{'parameter':58s} cos(HIP bf16, HIP fp32)  cos(oracle bf16-storage, HIP fp32)")
    for k, a in res["fp32"][2].items():
        if a.numel() < 64 or float(a.abs().max()) == 0.0:
            continue
        print(f"{k:58s} {cos(res['bf16'][2][k], a):8.4f} {cos(g_emu[k], a):8.4f}")

if os.environ.get("GRADS") == "2":
    # fp32 parity mode: every parameter gradient against the oracle's fp32 autograd at this size; with FP64=1 also
    # the oracle's own fp32 result against its fp64 one - the yardstick: hundreds of millions of ReLU / max-pool
    # decisions, some of them on a knife edge
    rel = lambda a, r: float((a.double() - r.double()).abs().max() / max(float(r.abs().max()), 1e-300))
    t0 = time.time()
    _, _, g32, _ = O.loss_and_grads(P, x0, x1, t)
    print(f"oracle fp32 fwd+bwd {time.time() - t0:.1f} s", flush=True)
    cols = {}
    if not NO_HIP:
        cols["HIP fp32 mode vs oracle fp32"] = {k: rel(res["fp32"][2][k], r) for k, r in g32.items()}
    if os.environ.get("FP64"):
        t0 = time.time()
        P64 = {k: (v.double() if v.is_floating_point() else v) for k, v in P.items()}
        te = O.time_embedding
        O.time_embedding = lambda tt, d: te(tt.float(), d).double()
        try:
            _, _, g64, _ = O.loss_and_grads(P64, x0.double(), x1.double(), t.double())
        finally:
            O.time_embedding = te
        print(f"oracle fp64 fwd+bwd {time.time() - t0:.1f} s", flush=True)
        cols["oracle fp32 vs oracle fp64"] = {k: rel(g32[k], g64[k]) for k in g32}
        if not NO_HIP:
            cols["HIP fp32 mode vs oracle fp64"] = {k: rel(res["fp32"][2][k], g64[k]) for k in g32}
    gmax = {k: float(r.abs().max()) for k, r in g32.items()}
    gscale = max(gmax.values())
    print("max-norm relative error per tensor (tensors >= 1e-3 of the largest gradient): " + " | ".join(cols))
    for k in sorted(g32, key=lambda k: -max(c[k] for c in cols.values())):
        if gmax[k] >= 1e-3 * gscale:
            print("  " + "  ".join(f"{c[k]:.3e}" for c in cols.values()) + f"  |ref|max {gmax[k]:.3e}  {k}")
    for name, c in cols.items():
        print(f"worst, {name}: {max(e for k, e in c.items() if gmax[k] >= 1e-3 * gscale):.3e}")
