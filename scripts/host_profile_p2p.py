"""cProfile of the host side of pix2pix G + D steps (where the ~15 us per launch go)."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stain2stain_amd import PatchGANDiscriminator, Pix2PixGenerator, Pix2PixTrainer
torch.manual_seed(0)
G, D = Pix2PixGenerator().cuda(), PatchGANDiscriminator().cuda()
tr = Pix2PixTrainer(G, D)
src = torch.rand(16, 3, 256, 256, device="cuda") * 2 - 1
tgt = torch.rand(16, 3, 256, 256, device="cuda") * 2 - 1
for _ in range(5):
    tr.step(src, tgt)
torch.cuda.synchronize()
import gc; gc.collect(); gc.disable()
pr = cProfile.Profile()
N = 20
pr.enable()
for _ in range(N):
    tr.step(src, tgt)
    torch.cuda.synchronize()
pr.disable()
for key in ("tottime", "cumtime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
    print(s.getvalue()[:6000])
