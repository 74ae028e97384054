"""Timing of the stem (3 -> 64 at 256x256) forward / weight-gradient kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stain2stain_amd import ops
B, dev, dt = 16, "cuda", torch.bfloat16
x = torch.rand(B, 3, 256, 256, device=dev) * 2 - 1
w = (torch.rand(64, 3, 3, 3, device=dev) - 0.5) * 0.2
b = torch.zeros(64, device=dev)
dy = torch.randn(B, 256, 256, 64, device=dev).to(dt)
dw, db = torch.empty_like(w), torch.empty_like(b)


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


print(f"stem fwd {timeit(lambda: ops.stem_fwd(x, w, b, dt, want_stats=True)):.1f} us | stem wgrad {timeit(lambda: ops.stem_wgrad(dy, x, dw, db)):.1f} us")

# fused head (1x1 conv 64 -> 3 + MSE + both backward passes)
xa = torch.randn(B, 256, 256, 64, device=dev).to(dt)
hw = (torch.rand(3, 64, 1, 1, device=dev) - 0.5) * 0.2
hb = torch.zeros(3, device=dev)
u = torch.randn(B, 3, 256, 256, device=dev)
hdw, hdb = torch.empty_like(hw), torch.empty_like(hb)
t = timeit(lambda: ops.head_loss_fused(xa, hw, hb, u, hdw, hdb))
print(f"head + loss {t:.1f} us = {2 * xa.numel() * 2 / t / 1e6:.2f} TB/s (env S2S_HEAD_BLOCKS={os.environ.get('S2S_HEAD_BLOCKS', '-')})")
loss, dx, _ = ops.head_loss_fused(xa, hw, hb, u, hdw, hdb)
print("head check", float(loss), float(dx.float().abs().sum()), float(hdw.abs().sum()), float(hdb.abs().sum()))
