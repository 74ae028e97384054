"""Smallest reproduction of a captured training step on a small network (diagnostics for tests/test_graph_step_gpu.py)."""
import faulthandler, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.enable()
import torch
from stain2stain_amd import CFMTrainer, FlowUNet

feats = [int(v) for v in os.environ.get("FEATS", "16,32,64").split(",")]
B, HW = int(os.environ.get("B", 4)), int(os.environ.get("HW", 64))
prec = os.environ.get("PREC", "bf16")
torch.manual_seed(1984)
net = FlowUNet(3, feats, 3, 32, precision=prec).to("cuda").train()
tr = CFMTrainer(net, lr=1e-3, graph=True)
tr.overlap_wgrad = os.environ.get("OVERLAP", "1") == "1"
g = torch.Generator().manual_seed(3)
for i in range(4):
    x0 = (torch.rand(B, 3, HW, HW, generator=g) * 2 - 1).cuda()
    x1 = (torch.rand(B, 3, HW, HW, generator=g) * 2 - 1).cuda()
    t = torch.rand(B, generator=g).cuda()
    print("step", i, flush=True)
    loss = tr.step(x0, x1, t)
    torch.cuda.synchronize()
    print("  loss", float(loss), flush=True)
print("ok")
