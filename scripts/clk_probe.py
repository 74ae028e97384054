"""Shader clock the conv kernel actually runs at under its own load (S2S_CONV_DBG=64 probes): random vs all-zero
operands, small-K vs large-K layers.  MI355X lowers the clock under MFMA load (DVFS), so the at-clock MFMA peak is
2500 TFLOP/s x MHz / 2400."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["S2S_CONV_DBG"] = "64"
import torch, numpy as np
from stain2stain_amd import ops, _native
dt = torch.bfloat16
lib = ctypes.CDLL(_native.LIB_PATH)
lib.s2s_debug_conv_clock.argtypes = [ctypes.c_void_p, ctypes.c_int]
for (H, cin, cout, zero) in [(64, 768, 256, False), (64, 768, 256, True), (256, 64, 64, False), (128, 384, 128, False)]:
    B = 16
    x = ((torch.rand(B, H, H, cin, device="cuda") * 2 - 1) * (0.0 if zero else 1.0)).to(dt)
    w = (torch.rand(cout, cin, 3, 3, device="cuda") - 0.5) * (0.0 if zero else 0.1)
    wf, wd = ops.pack_conv3x3(w, dt)
    y = torch.empty(B, H, H, cout, device="cuda", dtype=dt)
    bias = torch.zeros(cout, device="cuda")
    for _ in range(20):
        ops.conv3x3(x, None, wf, bias, cout, want_stats=True, out=y)
    torch.cuda.synchronize()
    n = 2048
    buf = (ctypes.c_long * (n * 4))()
    lib.s2s_debug_conv_clock(buf, n)
    t = np.frombuffer(buf, dtype=np.int64).reshape(n, 4).astype(np.float64)
    ok = (t[:, 3] > t[:, 2]) & (t[:, 1] > t[:, 0])
    mhz = ((t[ok, 1] - t[ok, 0]) / (t[ok, 3] - t[ok, 2])) * 100.0
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.conv3x3(x, None, wf, bias, cout, want_stats=True, out=y)
    e1.record(); torch.cuda.synchronize()
    tf = 2.0 * B * H * H * cout * 9 * cin / (e0.elapsed_time(e1) / 10 * 1e-3) / 1e12
    med = float(np.median(mhz))
    print((H, cin, cout), "zero" if zero else "rand", "shader clock MHz: median %.0f  p10 %.0f  p90 %.0f | %.0f TFLOP/s = %.0f%% of the "
          "at-clock MFMA peak (%.0f TFLOP/s)" % (med, np.percentile(mhz, 10), np.percentile(mhz, 90), tf, 100 * tf / (2500 * med / 2400), 2500 * med / 2400))
