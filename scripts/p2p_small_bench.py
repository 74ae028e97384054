"""Inner levels of the pix2pix generator (batch 16, ngf 64): the sample-complete launches of conv_small.hip against the
split-K convolution + single-launch InstanceNorm they replace, and the slab-free weight gradient against flat + fold."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stain2stain_amd import ops

DEV = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16


def timeit(fn, n=60):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def rnd(*shape):
    return torch.randn(*shape, device=DEV).to(torch.bfloat16)


print(f"batch {B}; us per call")
for name, ho, cin, cout in [("downs.4", 8, 512, 512), ("downs.5", 4, 512, 512), ("downs.6", 2, 512, 512), ("downs.7", 1, 512, 512)]:
    x = rnd(B, 2 * ho, 2 * ho, cin)
    w = torch.randn(cout, cin, 4, 4, device=DEV) / 90
    wf, wd = ops.pack_conv4x4_t(w, 2, torch.bfloat16)
    bias = torch.zeros(cout, device=DEV)
    norm = ho > 1

    def old():
        r = ops.conv4x4s2(x, wf, bias, cout, act=not norm, defer=norm)
        if norm:
            a = torch.empty((B, ho, ho, cout), dtype=torch.bfloat16, device=DEV)
            if isinstance(r, ops.SplitSum):
                ops.instnorm_lrelu_fwd2_split(r, 0.2, a)
            else:
                ops.instnorm_lrelu_fwd2(r, 0.2, a)

    def new():
        ops.convsm_fwd(1, x, wf, bias, cout, norm=norm, act=not norm, slope=0.2)

    g = rnd(B, ho, ho, cout)
    z = rnd(B, 2 * ho, 2 * ho, cin)
    st = torch.rand(4, B, cin, device=DEV)

    def old_b():
        d = ops.convT4x4s2(g, wd, None, cin, defer=True)
        ops.instnorm_lrelu_bwd2(d, None, z, st, 0.2)

    def new_b():
        ops.convsm_bwd(2, g, wd, cin, z=z, stats=st, g2=None, slope=0.2)

    gw = torch.empty(cout, cin, 4, 4, device=DEV)

    def wg():
        ops.convkxk_wgrad(g, x, gw, 2, x_plain=True)

    line = f"{name} conv {cin}->{cout} out {ho}^2: fwd+norm old {timeit(old):6.1f} new {timeit(new):6.1f}"
    if ho <= 4:
        line += f" | dgrad+norm bwd old {timeit(old_b):6.1f} new {timeit(new_b):6.1f}"
    line += f" | wgrad {timeit(wg):6.1f}"
    print(line)

for name, hi, cin, cout in [("ups.0", 1, 512, 512), ("ups.1", 2, 1024, 512), ("ups.2", 4, 1024, 512), ("ups.3", 8, 1024, 512)]:
    x = rnd(B, hi, hi, cin)
    w = torch.randn(cin, cout, 4, 4, device=DEV) / 90
    wf, wd = ops.pack_conv4x4_t(w, 2, torch.bfloat16)
    bias = torch.zeros(cout, device=DEV)

    def old():
        r = ops.convT4x4s2(x, wd, bias, cout, defer=True)
        a = torch.empty((B, 2 * hi, 2 * hi, cout), dtype=torch.bfloat16, device=DEV)
        if isinstance(r, ops.SplitSum):
            ops.instnorm_lrelu_fwd2_split(r, 0.0, a)
        else:
            ops.instnorm_lrelu_fwd2(r, 0.0, a)

    def new():
        ops.convsm_fwd(2, x, wd, bias, cout, norm=True, slope=0.0)

    g = rnd(B, 2 * hi, 2 * hi, cout)
    C = cin // 2
    z = rnd(B, hi, hi, C)
    st = torch.rand(4, B, C, device=DEV)

    def old_b():
        d = ops.conv4x4s2(g, wf, None, cin)
        if hi > 1:
            ops.instnorm_lrelu_bwd2(d[..., C:], None, z, st, 0.0)

    def new_b():
        if hi > 1:
            ops.convsm_bwd(1, g, wf, cin, z=z, stats=st, g2=None, slope=0.0, bwd_c0=C)
        else:
            ops.convsm_bwd(1, g, wf, cin, z=None, stats=None, g2=None, slope=0.0, bwd_c0=cin)

    gw = torch.empty(cin, cout, 4, 4, device=DEV)

    def wg():
        ops.convkxk_wgrad(x, g, gw, 2, x_plain=True)

    line = f"{name} convT {cin}->{cout} in {hi}^2: "
    if hi <= 4:
        line += f"fwd+norm old {timeit(old):6.1f} new {timeit(new):6.1f} | "
    line += f"dgrad+norm bwd old {timeit(old_b):6.1f} new {timeit(new_b):6.1f} | wgrad {timeit(wg):6.1f}"
    print(line)
