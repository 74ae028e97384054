"""N steps of the drop-in path (bench.py's dropin leg) for rocprofv3: python scripts/dropin_step.py [steps] [fused|torch]."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from functools import partial
import torch
from stain2stain_amd import ConditionalFlowMatchingModule, FlowUNet, FusedAdam

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cls = torch.optim.Adam if (len(sys.argv) > 2 and sys.argv[2] == "torch") else FusedAdam
dev = torch.device("cuda")
torch.manual_seed(1984)
net = FlowUNet(3, [64, 128, 256, 512, 1024], 3, 256).to(dev).train()
mod = ConditionalFlowMatchingModule(net, optimizer=partial(cls, lr=1e-4, weight_decay=1e-5))
opt = mod.configure_optimizers()["optimizer"]
g = torch.Generator().manual_seed(1984)
pool = [((torch.rand(16, 3, 256, 256, generator=g) * 2 - 1).to(dev), (torch.rand(16, 3, 256, 256, generator=g) * 2 - 1).to(dev)) for _ in range(4)]


def one(i):
    opt.zero_grad()
    loss = mod.training_step(pool[i % 4], i)
    loss.backward()
    opt.step()
    return loss


for i in range(3):
    one(i)
torch.cuda.synchronize()
import gc
gc.collect(); gc.disable()
t0 = time.perf_counter()
for i in range(steps):
    one(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{(t2 - t0) * 1e3 / steps:.3f} ms/step; host issue {(t1 - t0) * 1e3 / steps:.3f} ms/step")
