"""Per-step wall time of the first CFM steps after construction (is a short bench run still in a transient?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gc
from stain2stain_amd import CFMTrainer, FlowUNet
torch.manual_seed(0)
net = FlowUNet().cuda().train(); tr = CFMTrainer(net)
data = [(torch.rand(16, 3, 256, 256, device="cuda") * 2 - 1, torch.rand(16, 3, 256, 256, device="cuda") * 2 - 1, torch.rand(16, device="cuda")) for _ in range(4)]
gc.collect(); gc.disable()
ts = []
for i in range(40):
    torch.cuda.synchronize(); a = time.perf_counter()
    tr.step(*data[i % 4])
    torch.cuda.synchronize(); ts.append((time.perf_counter() - a) * 1e3)
print("per-step ms (synchronised each step):", " ".join(f"{t:.2f}" for t in ts))
for w, k in ((3, 10), (5, 20), (10, 60)):
    tr2 = None
    torch.cuda.synchronize(); 
    for i in range(w): tr.step(*data[i % 4])
    torch.cuda.synchronize(); a = time.perf_counter()
    for i in range(k): tr.step(*data[i % 4])
    torch.cuda.synchronize(); print(f"warmup {w} steps {k}: {(time.perf_counter() - a) / k * 1e3:.3f} ms/step")
