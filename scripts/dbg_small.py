import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, torch.nn.functional as F
import test_conv_small_gpu as T
from stain2stain_amd import ops
B, h, C, cout = 16, 2, 512, 512
g = torch.Generator().manual_seed(B * 7 + h)
skip = T._rb(torch.randn(B, C, h, h, generator=g)).requires_grad_(True)
zu = T._rb(torch.randn(B, C, h, h, generator=g)).requires_grad_(True)
w = T._rb(torch.randn(2 * C, cout, 4, 4, generator=g) / (2.0 * (2 * C) ** 0.5))
G = T._rb(torch.randn(B, cout, 2 * h, 2 * h, generator=g))
zn, mu, inv = T._inorm(zu)
(F.conv_transpose2d(torch.cat([skip, F.relu(zn)], 1), w, None, stride=2, padding=1) * G).sum().backward()
wf, _ = T._pack(w)
dz, plain = ops.convsm_bwd(1, T._nhwc(G), wf, 2 * C, z=T._nhwc(zu.detach()), stats=T._stats(zu.detach()), g2=None, slope=0.0, bwd_c0=C)
d = (T._nchw(dz) - zu.grad).abs()
ref = zu.grad
print("max ref", float(ref.abs().max()), "max err", float(d.max()), "plain err", T._rel(T._nchw(plain), skip.grad))
idx = (d > 0.02 * ref.abs().max()).nonzero()
print("bad elements", idx.shape[0], "of", d.numel())
for i in idx[:12]:
    n, c, y, x = [int(v) for v in i]
    print((n, c, y, x), "got", float(T._nchw(dz)[n, c, y, x]), "ref", float(ref[n, c, y, x]), "zn", zn[n, c].flatten().tolist(), "inv", float(inv[n, c]))
