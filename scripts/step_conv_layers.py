"""Per-launch times of the fwd / dgrad conv3x3 calls INSIDE the CFM training step (event-bracketed, one stream), in call
order -- what a layer costs with the step's real operands (strides of the concat buffers, cache state left by the
neighbouring kernels), which the stand-alone scripts/conv_bench.py does not reproduce.  S2S_CONV_STAGE=0/1/2 to compare."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stain2stain_amd import ops, CFMTrainer, FlowUNet

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = FlowUNet(3, [64, 128, 256, 512, 1024], 3, 256).to(dev).train()
B = int(os.environ.get("B", 16))
tr = CFMTrainer(net, lr=1e-4, weight_decay=1e-5)
g = torch.Generator().manual_seed(1)
pool = [((torch.rand(B, 3, 256, 256, generator=g) * 2 - 1).to(dev), (torch.rand(B, 3, 256, 256, generator=g) * 2 - 1).to(dev)) for _ in range(2)]
ts = [torch.rand(B, generator=g).to(dev) for _ in range(16)]
for i in range(6):
    tr.step(*pool[i % 2], ts[i])
torch.cuda.synchronize()
tr.overlap_wgrad = False
rows = {}
N = 6
for i in range(N):
    ops.profile_start(("conv3x3_mfma", "conv3x3_wgrad_mfma"))
    tr.step(*pool[i % 2], ts[6 + i])
    prof = ops.profile_stop()
    torch.cuda.synchronize()
    k = 0
    for name, work, e0, e1 in prof:
        rows.setdefault((k, name, work), []).append(e0.elapsed_time(e1) * 1e3)
        k += 1
tot = {}
for (k, name, work), v in sorted(rows.items()):
    v = sorted(v)[len(v) // 2]
    tot[name.split("@")[0]] = tot.get(name.split("@")[0], 0.0) + v
    print(f"{k:3d} {name:34s} {work / 1e9:7.1f} GF {v:8.1f} us {work / v / 1e6:7.0f} TF")
print({k: round(v, 1) for k, v in tot.items()})
tr.close()
