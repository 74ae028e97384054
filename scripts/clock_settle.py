"""the two clock readings for the SAME launches.  For the four shapes of clk_probe.py the conv kernel
is launched N times back to back with the in-kernel probe on (S2S_CONV_DBG=64: every workgroup stamps s_memtime and the
100 MHz s_memrealtime at entry and exit); after the last launch the stamps are read back and printed with the number of
conv dispatches issued so far.  Run once plainly (the chip under the kernel's own sustained load) and once under
`rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv`: scripts/clock_summary.py then puts
GRBM_GUI_ACTIVE / 8 / duration of exactly that dispatch next to the in-kernel figure (profiles/r04_clock.json)."""
import os, sys, ctypes, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["S2S_CONV_DBG"] = "64"
import torch, numpy as np
from stain2stain_amd import ops, _native
dt = torch.bfloat16
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
lib = ctypes.CDLL(_native.LIB_PATH)
lib.s2s_debug_conv_clock.argtypes = [ctypes.c_void_p, ctypes.c_int]
issued = 0
for (H, cin, cout, zero) in [(64, 768, 256, False), (64, 768, 256, True), (256, 64, 64, False), (128, 384, 128, False)]:
    B = 16
    x = ((torch.rand(B, H, H, cin, device="cuda") * 2 - 1) * (0.0 if zero else 1.0)).to(dt)
    w = (torch.rand(cout, cin, 3, 3, device="cuda") - 0.5) * (0.0 if zero else 0.1)
    wf, wd = ops.pack_conv3x3(w, dt)
    y = torch.empty(B, H, H, cout, device="cuda", dtype=dt)
    bias = torch.zeros(cout, device="cuda")
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N):
        ops.conv3x3(x, None, wf, bias, cout, want_stats=True, out=y)
    e1.record(); torch.cuda.synchronize()
    issued += N
    n = 2048
    buf = (ctypes.c_long * (n * 4))()
    lib.s2s_debug_conv_clock(buf, n)
    t = np.frombuffer(buf, dtype=np.int64).reshape(n, 4).astype(np.float64)
    ok = (t[:, 3] > t[:, 2]) & (t[:, 1] > t[:, 0])
    mhz = ((t[ok, 1] - t[ok, 0]) / (t[ok, 3] - t[ok, 2])) * 100.0
    ms = e0.elapsed_time(e1) / N
    print(json.dumps({"shape": [H, cin, cout], "operands": "zero" if zero else "random", "launches": N,
                      "conv_dispatch_index_of_last_launch": issued, "ms_per_launch_back_to_back": round(ms, 4),
                      "seconds_of_load": round(ms * N / 1e3, 2),
                      "tflops": round(2.0 * B * H * H * cout * 9 * cin / (ms * 1e-3) / 1e12, 1),
                      "memtime_mhz_median": round(float(np.median(mhz)), 0), "memtime_mhz_p10": round(float(np.percentile(mhz, 10)), 0),
                      "memtime_mhz_p90": round(float(np.percentile(mhz, 90)), 0), "workgroups_stamped": int(ok.sum())}), flush=True)
