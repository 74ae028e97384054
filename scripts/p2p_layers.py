#!/usr/bin/env python3
"""Per-layer rates of the pix2pix networks at the bench configuration (batch 16, 256x256, bf16): forward, data gradient
and weight gradient of every 4x4 layer, HIP-event timed (20 launches each after 5 warm-ups), FLOP on the padded channel
counts the kernels execute.  Run on the GPU box:  python scripts/p2p_layers.py [batch]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from stain2stain_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev, dt = "cuda", torch.bfloat16


def timed(fn, n=20, w=5):
    for _ in range(w):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / n


def row(name, flop, tf, td, tw):
    f = lambda t: f"{t * 1e6:7.1f} us {flop / t / 1e12:7.1f} TF" if t else "      -            "
    print(f"{name:34s} {flop / 1e9:7.2f} GF | fwd {f(tf)} | dgrad {f(td)} | wgrad {f(tw)}")


ch = [64, 128, 256, 512, 512, 512, 512, 512]
print(f"batch {B}, bf16; generator down path (conv 4x4 s2), input side = 256 >> i")
for i in range(8):
    cin, cout, hin = (8 if i == 0 else ch[i - 1]), ch[i], 256 >> i
    x = torch.randn(B, hin, hin, cin, device=dev).to(dt)
    w = torch.randn(cout, cin, 4, 4, device=dev) * 0.05
    wf, wd = ops.pack_conv4x4_t(w, 2, dt)
    g = torch.randn(B, hin // 2, hin // 2, cout, device=dev).to(dt)
    gw = torch.empty_like(w)
    flop = 2.0 * B * (hin // 2) ** 2 * cout * 16 * cin
    tf = timed(lambda: ops.conv4x4s2(x, wf, None, cout))
    td = timed(lambda: ops.convT4x4s2(g, wd, None, cin)) if cin % 64 == 0 else None
    tw = timed(lambda: ops.convkxk_wgrad(g, x, gw, 2, x_plain=True))
    row(f"G.downs.{i} {cin}->{cout} @{hin // 2}^2", flop, tf, td, tw)
print("generator up path (transposed conv 4x4 s2), output side = 2 h")
for j in range(8):
    i = 7 - j
    cin = ch[i] if j == 0 else 2 * ch[i]
    cout = 8 if i == 0 else ch[i - 1]
    h = 256 >> (i + 1)
    x = torch.randn(B, h, h, cin, device=dev).to(dt)
    w = torch.randn(cin, cout, 4, 4, device=dev) * 0.05
    wf, wd = ops.pack_conv4x4_t(w, 2, dt)
    g = torch.randn(B, 2 * h, 2 * h, cout, device=dev).to(dt)
    gw = torch.empty_like(w)
    flop = 2.0 * B * (2 * h) ** 2 * cout * 4 * cin
    if cout % 64 == 0:
        tf = timed(lambda: ops.convT4x4s2(x, wd, None, cout))
    else:
        tf = timed(lambda: ops.depth_to_space_unpad1_t(ops.convkxk(x, wd, None, 4 * cout, 2, 1)))
    td = timed(lambda: ops.conv4x4s2(g, wf, None, cin))
    tw = timed(lambda: ops.convkxk_wgrad(x, g, gw, 2, x_plain=True))
    row(f"G.ups.{j} {cin}->{cout} @{2 * h}^2", flop, tf, td, tw)
print(f"discriminator (update pass: batch {2 * B})")
for name, cin, cout, hin, st in (("c1", 8, 64, 256, 2), ("c2", 64, 128, 128, 2), ("c3", 128, 256, 64, 2),
                                 ("c4", 256, 512, 32, 1), ("c5", 512, 8, 31, 1)):
    N = 2 * B
    x = torch.randn(N, hin, hin, cin, device=dev).to(dt)
    w = torch.randn(cout, cin, 4, 4, device=dev) * 0.05
    wf, wd = ops.pack_conv4x4_t(w, st, dt)
    ho = hin // 2 if st == 2 else hin - 1
    g = torch.randn(N, ho, ho, cout, device=dev).to(dt)
    gw = torch.empty_like(w)
    flop = 2.0 * N * ho * ho * cout * 16 * cin
    if st == 2:
        tf = timed(lambda: ops.conv4x4s2(x, wf, None, cout))
        td = timed(lambda: ops.convT4x4s2(g, wd, None, cin)) if cin % 64 == 0 else None
        tw = timed(lambda: ops.convkxk_wgrad(g, x, gw, 2, x_plain=True))
    else:
        tf = timed(lambda: ops.convkxk(x, wf, None, cout, 4, 1))
        td = timed(lambda: ops.convkxk(g, wd, None, cin, 4, 2))
        tw = timed(lambda: ops.convkxk_wgrad(g, x, gw, 4))
    row(f"D.{name} {cin}->{cout} @{ho}^2", flop, tf, td, tw)
