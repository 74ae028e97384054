"""Would the stem (3 -> 64 at 256 x 256) be faster through the generic LDS-DMA 3x3 kernel on an 8-channel NHWC bf16 copy of
the image?  (stem_mfma_kernel: 106 us; its output alone is 134 MB = 28 us at 4.8 TB/s.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stain2stain_amd import ops
B, dev, dt = 16, "cuda", torch.bfloat16
torch.manual_seed(0)
x = torch.rand(B, 3, 256, 256, device=dev) * 2 - 1
w = (torch.rand(64, 3, 3, 3, device=dev) - 0.5) * 0.2
b = torch.zeros(64, device=dev)


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


w8 = torch.zeros(64, 8, 3, 3, device=dev); w8[:, :3] = w
wf, _ = ops.pack_conv3x3(w8, dt, want_dgrad=False)
x8 = torch.empty(B, 256, 256, 8, device=dev, dtype=dt)
pack = lambda: ops.p2p_pack_input(x, None, x8)
pack()
y_ref, st_ref = ops.stem_fwd(x, w, None, dt, want_stats=True)
y, st = ops.conv3x3(x8, None, wf, None, 64, want_stats=True)
print("max |y - y_stem|", float((y.float() - y_ref.float()).abs().max()), "of", float(y_ref.float().abs().max()))
print("stats rows", st.shape, st_ref.shape, "sum diff", float((st.sum(2) - st_ref.sum(2)).abs().max()), "of", float(st_ref.sum(2).abs().max()))
print(f"stem kernel {timeit(lambda: ops.stem_fwd(x, w, None, dt, want_stats=True)):.1f} us | pack {timeit(pack):.1f} us | "
      f"generic conv on 8 ch {timeit(lambda: ops.conv3x3(x8, None, wf, None, 64, want_stats=True)):.1f} us")
