"""Run one conv layer shape a few times (for rocprofv3 --pmc): python scripts/one_layer.py H cin cout which"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stain2stain_amd import ops
H, cin, cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]); which = sys.argv[4]
B = 16; dt = torch.bfloat16; dev = "cuda"
x = (torch.rand(B, H, H, cin, device=dev) * 2 - 1).to(dt)
w = (torch.rand(cout, cin, 3, 3, device=dev) - 0.5) * 0.1
wf, wd = ops.pack_conv3x3(w, dt)
dy = (torch.rand(B, H, H, cout, device=dev) * 2 - 1).to(dt)
y = torch.empty(B, H, H, cout, device=dev, dtype=dt)
grad = torch.empty(cout, cin, 3, 3, device=dev)
for _ in range(5):
    if which == "fwd":
        ops.conv3x3(x, None, wf, None, cout, want_stats=True, out=y)
    else:
        ops.conv3x3_wgrad(dy, x, None, grad)
torch.cuda.synchronize()
