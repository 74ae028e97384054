#!/usr/bin/env python3
"""Per-parameter gradient error of the pix2pix engine's fp32 mode against the torch-layer oracle (fp64 autograd) at the
bench configuration's network sizes, batch B (default 1).  Diagnostic for tests/test_pix2pix_engine_gpu.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from test_pix2pix_engine_gpu import _engine_grads, _oracle_eval  # noqa: E402
from oracle import pix2pix_oracle as O  # noqa: E402
from stain2stain_amd import PatchGANDiscriminator, Pix2PixGenerator, Pix2PixTrainer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 300
torch.manual_seed(1984)
G, D = Pix2PixGenerator(), PatchGANDiscriminator()
Go, Do = O.OracleGenerator(), O.OracleDiscriminator()
Go.load_state_dict(G.state_dict()); Do.load_state_dict(D.state_dict())
g = torch.Generator().manual_seed(seed)
src, tgt = torch.rand(B, 3, 256, 256, generator=g) * 2 - 1, torch.rand(B, 3, 256, 256, generator=g) * 2 - 1
torch.set_num_threads(min(16, os.cpu_count() or 8))
fake_o, ld_o, lg_o, gref = _oracle_eval(Go, Do, src, tgt, torch.float64)
g32 = _oracle_eval(Go, Do, src, tgt, torch.float32)[3]
tr = Pix2PixTrainer(G.cuda(), D.cuda(), precision="fp32")
losses, fake = tr.losses_and_grads(src.cuda(), tgt.cuda(), update=False, want_fake=True)
got = _engine_grads(tr)
print("fake", float((fake.cpu().double() - fake_o).abs().max() / fake_o.abs().max()), tr.loss_values(losses), ld_o, lg_o)
sc = {"G": max(float(v.abs().max()) for k, v in gref.items() if k[0] == "G"), "D": max(float(v.abs().max()) for k, v in gref.items() if k[0] == "D")}
for k, r in gref.items():
    den = max(float(r.abs().max()), 1e-3 * sc[k[0]])
    l2 = lambda a: float((a - r).norm() / max(float(r.norm()), 1e-30))
    print(f"{k:22s} |g| {float(r.abs().max()):.3e}  max-norm: engine {float((got[k] - r).abs().max()) / den:.2e} oracle-fp32 "
          f"{float((g32[k] - r).abs().max()) / den:.2e}   L2: engine {l2(got[k]):.2e} oracle-fp32 {l2(g32[k]):.2e}")
