import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from stain2stain_amd import ops
torch.manual_seed(0)
for (B, h, C) in [(16, 128, 128), (16, 64, 256), (16, 32, 512), (16, 16, 1024)]:
    dy = torch.randn(B, 2 * h, 2 * h, C, device="cuda").to(torch.bfloat16)
    outs = []
    for _ in range(3): dx = ops.upsample2x_bwd(dy, h, h)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): dx = ops.upsample2x_bwd(dy, h, h)
    e1.record(); torch.cuda.synchronize()
    print(os.environ.get("S2S_UP_BWD_WIN", "default"), (B, h, C), "%.1f us" % (e0.elapsed_time(e1) / 20 * 1e3), float(dx.float().abs().sum()))
    torch.save(dx.cpu(), f"/tmp/up_{os.environ.get('S2S_UP_BWD_WIN','d')}_{h}.pt")
