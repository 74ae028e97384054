"""Timing of the BatchNorm forward-finalize / backward passes at the headline layer shapes (batch 16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stain2stain_amd import ops

B, dev, dt = 16, "cuda", torch.bfloat16
SHAPES = [(256, 64), (128, 128), (64, 256), (32, 512), (16, 1024)]


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


for H, C in SHAPES:
    x = torch.randn(B, H, H, C, device=dev).to(dt)
    g = torch.randn(B, H, H, C, device=dev).to(dt)
    gp = torch.randn(B, H // 2, H // 2, C, device=dev).to(dt)
    gamma, beta = torch.rand(C, device=dev) + 0.5, torch.rand(C, device=dev)
    rm, rv, nbt = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros((), dtype=torch.long, device=dev)
    nblk = B * (H // 8) * max(H // 32, 1)
    stat = torch.rand(2, C, nblk, device=dev)
    st = ops.bn_finalize(stat.clone(), B * H * H, gamma, beta, rm, rv, nbt)
    dg, db, dbias = torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    t_fin = timeit(lambda: ops.bn_finalize(stat, B * H * H, gamma, beta, rm, rv, nbt))
    t_apply = timeit(lambda: ops.bn_relu_apply(x, st[2], st[3]))
    t_flat = timeit(lambda: ops.bn_relu_bwd(g, None, x, st, gamma, dg, db, dbias))
    t_win = timeit(lambda: ops.bn_relu_bwd(g, gp, x, st, gamma, dg, db, dbias))
    gb = B * H * H * C * 2 / 1e9
    print(f"H{H:4d} C{C:5d} rows {nblk:5d} | finalize {t_fin:6.1f} us | apply {t_apply:6.1f} us ({2*gb/t_apply*1e3:5.2f} TB/s)"
          f" | bwd flat {t_flat:6.1f} us ({5*gb/t_flat*1e3:5.2f} TB/s) | bwd pool {t_win:6.1f} us", flush=True)

print("upsample x2 (decoder shapes)")
for Hin, C in [(16, 1024), (32, 512), (64, 256), (128, 128)]:
    x = torch.randn(B, Hin, Hin, C, device=dev).to(dt)
    cat = torch.empty(B, 2 * Hin, 2 * Hin, 2 * C, device=dev, dtype=dt)
    out = cat[..., C:]
    dy = torch.randn(B, 2 * Hin, 2 * Hin, C, device=dev).to(dt)
    bias = torch.randn(B, C, device=dev)
    t_f = timeit(lambda: ops.upsample2x_fwd(x, out))
    t_fb = timeit(lambda: ops.upsample2x_fwd(x, out, bias))
    t_b = timeit(lambda: ops.upsample2x_bwd(dy, Hin, Hin))
    gb = B * Hin * Hin * C * 2 * 5 / 1e9
    print(f"Hin{Hin:4d} C{C:5d} | fwd {t_f:6.1f} us ({gb/t_f*1e3:5.2f} TB/s) | fwd+bias {t_fb:6.1f} us | bwd {t_b:6.1f} us ({gb/t_b*1e3:5.2f} TB/s)", flush=True)

print("InstanceNorm + LeakyReLU (pix2pix generator levels, batch 16)")
for H, C in [(128, 64), (64, 128), (32, 256), (16, 512)]:
    x = torch.randn(B, H, H, C, device=dev).to(dt)
    g = torch.randn(B, H, H, C, device=dev).to(dt)
    y, st = ops.instnorm_lrelu_fwd(x, None, None)
    t_f = timeit(lambda: ops.instnorm_lrelu_fwd(x, None, None))
    t_b = timeit(lambda: ops.instnorm_lrelu_bwd(g, x, st))
    gb = B * H * H * C * 2 / 1e9
    print(f"H{H:4d} C{C:5d} | fwd {t_f:6.1f} us ({3*gb/t_f*1e3:5.2f} TB/s) | bwd {t_b:6.1f} us ({5*gb/t_b*1e3:5.2f} TB/s)", flush=True)
