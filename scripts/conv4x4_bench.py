"""Time of the 2x2-tap MFMA kernel on the pix2pix generator's 4x4 stride-2 layers (batch 16, 256x256 input): kernel
only, forward and data-gradient / transposed form, and the forward including the torch space-to-depth copy."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stain2stain_amd import ops, pix2pix as P

B, dev, dt = 16, "cuda", torch.bfloat16
LAYERS = [(256, 8, 64), (128, 64, 128), (64, 128, 256), (32, 256, 512), (16, 512, 512), (8, 512, 512)]   # H_in, Cin, Cout


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


for H, cin, cout in LAYERS:
    x = (torch.rand(B, H, H, cin, device=dev) * 2 - 1).to(dt)
    dy = (torch.rand(B, H // 2, H // 2, cout, device=dev) - 0.5).to(dt)
    w = (torch.rand(cout, cin, 4, 4, device=dev) - 0.5) * 0.05
    wf, wd = P.pack_conv4x4_s2(w)
    xs = P.space_to_depth_pad1(x)
    gf = 2.0 * B * (H // 2) ** 2 * cout * 16 * cin / 1e9
    t_k = timeit(lambda: ops.conv2x2(xs, wf, None, cout, 0))
    t_d = timeit(lambda: ops.conv2x2(dy, wd, None, 4 * cin, 1))
    t_all = timeit(lambda: P.conv4x4_s2(x, wf, None, cout))
    print(f"H{H:4d} {cin:4d}->{cout:4d} {gf:7.1f} GF | fwd kernel {t_k:7.1f} us {gf/t_k*1e3:6.0f} TF | "
          f"dgrad/transposed kernel {t_d:7.1f} us {gf/t_d*1e3:6.0f} TF | fwd incl. torch s2d {t_all:7.1f} us", flush=True)
