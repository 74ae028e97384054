"""VERDICT r3 item 4, the A/B that decides it: the forward conv of the two probe layers (64 -> 64 at 256^2, 512 -> 512 at
32^2) with and without an in-LDS affine + ReLU pass over every halo chunk (ablate build, S2S_CONV_DBG=1024; timing only),
against the bn_relu_apply launch such a consumer-side BatchNorm would remove.  Run as two processes on one box:
    S2S_CONV_PERS=0 python scripts/bn_in_conv_ab.py            (baseline)
    S2S_CONV_PERS=0 S2S_CONV_DBG=1024 python scripts/bn_in_conv_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ablate_lib  # noqa: F401  (first: loads the -DS2S_ABLATE library)
import torch
from stain2stain_amd import ops

dt, dev, B = torch.bfloat16, "cuda", 16


def timeit(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for H, cin, cout in [(256, 64, 64), (32, 512, 512), (64, 256, 256), (128, 128, 128)]:
    x = (torch.rand(B, H, H, cin, device=dev) * 2 - 1).to(dt)
    w = (torch.rand(cout, cin, 3, 3, device=dev) - 0.5) * 0.1
    wf, _ = ops.pack_conv3x3(w, dt)
    y = torch.empty(B, H, H, cout, device=dev, dtype=dt)
    sc = torch.ones(cout, device=dev); sh = torch.zeros(cout, device=dev)
    tc = timeit(lambda: ops.conv3x3(x, None, wf, None, cout, want_stats=True, out=y))
    ta = timeit(lambda: ops.bn_relu_apply(y, sc, sh))
    print(f"DBG={os.environ.get('S2S_CONV_DBG', '0'):>5} H{H:4d} {cin:4d}->{cout:4d}: conv {tc:7.1f} us | bn_relu_apply of its output {ta:6.1f} us")
