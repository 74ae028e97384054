"""Timing ablations of conv3x3_stage_kernel (default 64 -> 64 at 256^2, batch 16; argv: cout cin H) on the -DS2S_ABLATE
library: S2S_CONV_DBG bit 1 = no weight DMA, 2 = no fragment reads / MFMAs, 4 = no halo DMA, 8 = no global stores,
16 = no epilogue.  Results are wrong by construction."""
import ablate_lib  # noqa: F401  (first: builds and loads the ablation library)
import os
import sys
import torch
from stain2stain_amd import ops

B = 16
cout = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cin = int(sys.argv[2]) if len(sys.argv) > 2 else 64
H = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dt = torch.bfloat16
x = (torch.rand(B, H, H, cin, device="cuda") * 2 - 1).to(dt)
w = (torch.rand(cout, cin, 3, 3, device="cuda") - 0.5) * 0.1
wf, _ = ops.pack_conv3x3(w, dt)
y = torch.empty(B, H, H, cout, device="cuda", dtype=dt)
for stats in (False, True):
    for _ in range(3):
        ops.conv3x3(x, None, wf, None, cout, want_stats=stats, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.conv3x3(x, None, wf, None, cout, want_stats=stats, out=y)
    e1.record()
    torch.cuda.synchronize()
    print(f"DBG={os.environ.get('S2S_CONV_DBG', '0'):>3} {cin}->{cout}@{H} stats={int(stats)}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us", flush=True)
