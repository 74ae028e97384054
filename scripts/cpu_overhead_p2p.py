"""Host enqueue time of one pix2pix G + D step (Python + ctypes launches) against its total time at batch 16."""
import sys, os, time
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
cpu = []
for _ in range(20):
    torch.cuda.synchronize(); a = time.perf_counter(); tr.step(src, tgt); b = time.perf_counter(); torch.cuda.synchronize(); c = time.perf_counter()
    cpu.append((b - a, c - a))
print("pix2pix: cpu issue ms %.3f  total ms %.3f" % (sum(x[0] for x in cpu) / len(cpu) * 1e3, sum(x[1] for x in cpu) / len(cpu) * 1e3))
a = time.perf_counter()
for _ in range(50):
    tr.step(src, tgt)
torch.cuda.synchronize()
print("pix2pix: back-to-back ms/step %.3f" % ((time.perf_counter() - a) / 50 * 1e3))
