"""Host enqueue time of one training step (Python + ctypes launches) against its total time, at batch 16 and in the
launch-bound regime of batch 2: the headline step is GPU-bound (enqueue time < GPU time)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stain2stain_amd import CFMTrainer, FlowUNet
torch.manual_seed(0)
net = FlowUNet().cuda().train(); tr = CFMTrainer(net)
x0 = torch.rand(16,3,256,256,device='cuda')*2-1; x1 = torch.rand(16,3,256,256,device='cuda')*2-1; t = torch.rand(16,device='cuda')
for _ in range(3): tr.step(x0,x1,t)
torch.cuda.synchronize()
cpu=[]; 
for _ in range(10):
    torch.cuda.synchronize(); a=time.perf_counter(); tr.step(x0,x1,t); b=time.perf_counter(); torch.cuda.synchronize(); c=time.perf_counter()
    cpu.append((b-a, c-a))
print('cpu issue ms', sum(x[0] for x in cpu)/len(cpu)*1e3, ' total ms', sum(x[1] for x in cpu)/len(cpu)*1e3)
# small batch: launch-bound regime
x0s=x0[:2].contiguous(); x1s=x1[:2].contiguous(); ts=t[:2].contiguous()
for _ in range(3): tr.step(x0s,x1s,ts)
torch.cuda.synchronize(); a=time.perf_counter()
for _ in range(10): tr.step(x0s,x1s,ts)
torch.cuda.synchronize(); print('batch 2: ms/step', (time.perf_counter()-a)/10*1e3)
