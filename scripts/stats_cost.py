"""What do the BatchNorm partial-sum rows cost the producing convolution?  (one row per (tile, wave row): 4096 tiles x 4
rows x 2 x 64 channels at 256 x 256 -- 2.1 M four-byte stores, each to its own cache line of the channel-major buffer)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stain2stain_amd import ops
B, dev, dt = 16, "cuda", torch.bfloat16


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


x = torch.rand(B, 3, 256, 256, device=dev) * 2 - 1
w = (torch.rand(64, 3, 3, 3, device=dev) - 0.5) * 0.2
print(f"stem: stats {timeit(lambda: ops.stem_fwd(x, w, None, dt, want_stats=True)):.1f} us, no stats {timeit(lambda: ops.stem_fwd(x, w, None, dt, want_stats=False)):.1f} us")
for (H, cin, cout) in [(256, 64, 64), (256, 192, 64), (128, 128, 128), (64, 256, 256)]:
    xa = (torch.rand(B, H, H, cin, device=dev) * 2 - 1).to(dt)
    ww = (torch.rand(cout, cin, 3, 3, device=dev) - 0.5) * 0.1
    wf, _ = ops.pack_conv3x3(ww, dt, want_dgrad=False)
    y = torch.empty(B, H, H, cout, device=dev, dtype=dt)
    a = timeit(lambda: ops.conv3x3(xa, None, wf, None, cout, want_stats=True, out=y))
    b = timeit(lambda: ops.conv3x3(xa, None, wf, None, cout, want_stats=False, out=y))
    print(f"conv {cin}->{cout} @{H}: stats {a:.1f} us, no stats {b:.1f} us")
