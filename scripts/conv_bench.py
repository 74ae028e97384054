"""Per-layer-shape timing of the MFMA conv kernels at the headline configuration (batch 16, 256x256)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stain2stain_amd import ops

B = int(os.environ.get("B", 16))
# (H, c0, c1, cout)
FWD = [(256, 64, 0, 64), (128, 64, 0, 128), (128, 128, 0, 128), (64, 128, 0, 256), (64, 256, 0, 256),
       (32, 256, 0, 512), (32, 512, 0, 512), (16, 512, 0, 1024), (16, 1024, 0, 1024),
       (32, 512, 1024, 512), (64, 256, 512, 256), (128, 128, 256, 128), (256, 64, 128, 64)]
dt = torch.bfloat16
dev = "cuda"


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


which = sys.argv[1] if len(sys.argv) > 1 else "all"
tot = {"fwd": [0, 0], "dgrad": [0, 0], "wgrad": [0, 0]}
if os.environ.get("LAYERS"):
    FWD = [FWD[int(i)] for i in os.environ["LAYERS"].split(",")]
for (H, c0, c1, cout) in FWD:
    cin = c0 + c1
    x = ((torch.rand(B, H, H, cin, device=dev) * 2 - 1) * float(os.environ.get("ZERO", "1") != "0")).to(dt)
    w = (torch.rand(cout, cin, 3, 3, device=dev) - 0.5) * 0.1 * float(os.environ.get("ZERO", "1") != "0")
    wf, wd = ops.pack_conv3x3(w, dt)
    dy = (torch.rand(B, H, H, cout, device=dev) * 2 - 1).to(dt)
    bias = None if os.environ.get("NOBIAS") else torch.zeros(cout, device=dev)      # (the training path convolves without it)
    gflop = 2.0 * B * H * H * cout * 9 * cin / 1e9
    x0 = x[..., :c0]; x1 = x[..., c0:] if c1 else None
    y = torch.empty(B, H, H, cout, device=dev, dtype=dt)
    dx = torch.empty(B, H, H, cin, device=dev, dtype=dt)
    grad = torch.empty(cout, cin, 3, 3, device=dev)
    row = f"H{H:4d} cin{cin:5d} cout{cout:5d} {gflop:8.1f} GF |"
    if which in ("all", "fwd", "fd"):
        t = timeit(lambda: ops.conv3x3(x0, x1, wf, bias, cout, want_stats=True, out=y))
        row += f" fwd {t*1e3:7.1f} us {gflop/t:7.0f} TF |"; tot["fwd"][0] += gflop; tot["fwd"][1] += t
    if which in ("all", "dgrad", "fd"):
        t = timeit(lambda: ops.conv3x3(dy, None, wd, None, cin, out=dx))
        row += f" dgrad {t*1e3:7.1f} us {gflop/t:7.0f} TF |"; tot["dgrad"][0] += gflop; tot["dgrad"][1] += t
    if which in ("all", "wgrad"):
        t = timeit(lambda: ops.conv3x3_wgrad(dy, x0, x1, grad))
        row += f" wgrad {t*1e3:7.1f} us {gflop/t:7.0f} TF"; tot["wgrad"][0] += gflop; tot["wgrad"][1] += t
    if which in ("all", "gemm"):
        # calibration only: the library GEMM of the same M x N x K (what an im2col lowering would run, without
        # the im2col traffic) -- an upper reference for what this GPU sustains on this shape under load
        a = (torch.rand(B * H * H, 9 * cin, device=dev) * 2 - 1).to(dt)
        bm = ((torch.rand(9 * cin, cout, device=dev) - 0.5) * 0.1).to(dt)
        t = timeit(lambda: torch.mm(a, bm))
        row += f" | lib gemm {t*1e3:7.1f} us {gflop/t:7.0f} TF"
        tot.setdefault("gemm", [0, 0]); tot["gemm"][0] += gflop; tot["gemm"][1] += t
        del a, bm
    print(row, flush=True)
for k, (g, t) in tot.items():
    if t:
        print(f"{k}: {g:.0f} GF in {t:.3f} ms = {g/t:.0f} TFLOP/s")
