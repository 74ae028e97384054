"""Per-parameter gradient error of the HIP path against the golden tiny step (fp32 mode): the table behind the
6.8e-6 figure in DESIGN.md section 3.3."""
import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import torch
from conftest import load_golden, sub
from stain2stain_amd import CFMTrainer, FlowUNet
G=load_golden('tiny_step.npz')
for prec in ('bf16','fp32'):
    net=FlowUNet(3,[16,32],3,32,precision=prec); net.load_state_dict(sub(G,'init/')); net=net.cuda().train()
    tr=CFMTrainer(net,lr=1e-4,weight_decay=1e-5)
    x0,x1,t=(G[f'step0/{k}'].cuda() for k in ('x0','x1','t'))
    loss,v=tr.forward_backward(x0,x1,t)
    ref=sub(G,'step0/grad/')
    got={"encoder."+k:p.grad for k,p in net.encoder.named_parameters()}
    got.update({"flow_decoder."+k:p.grad for k,p in net.flow_decoder.named_parameters()})
    rows=[]
    for k,r in ref.items():
        e=float((got[k].cpu()-r).abs().max()); rows.append((e/max(float(r.abs().max()),1e-30), float(r.abs().max()), k))
    rows.sort(reverse=True)
    print(prec)
    for r in rows[:12]: print('  %.3e  |ref|max %.3e  %s'%r)
