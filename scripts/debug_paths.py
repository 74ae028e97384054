"""Walks the golden tiny step (post-step-0 parameters) layer by layer through the CPU oracle and the HIP engine and
prints, per activation, the relative error and the number of ReLU decisions that differ (the knife-edge analysis
quoted in tests/test_e2e_gpu.py)."""
import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import torch
from conftest import load_golden, sub
from stain2stain_amd import FlowUNet, ops, engine
from oracle import unet_oracle as O
G=load_golden('tiny_step.npz')
x0,x1,t=(G[f'step1/{k}'] for k in ('x0','x1','t'))
P=sub(G,'step0/after/')
net=FlowUNet(3,[16,32],3,32,precision='fp32'); net.load_state_dict(P); net=net.cuda().train()
xt,ut=O.cfm_sample(x0,x1,t)
b,sk=O.encoder_forward(xt,P,True)
# oracle decoder pieces
temb=O.time_embedding(t,32)
h=O.linear(temb,P,'flow_decoder.time_mlp.0'); h=h*torch.sigmoid(h); h=O.linear(h,P,'flow_decoder.time_mlp.2'); tb=O.linear(h,P,'flow_decoder.time_proj')
x=b+tb[:,:,None,None]
up=O.upsample2x_bilinear_ac(x)
cat=torch.cat([sk[0],up],1)
a1=O.conv_bn_relu(cat,P,'flow_decoder.ups.0.conv.double_conv.0','flow_decoder.ups.0.conv.double_conv.1',True,None)
a2=O.conv_bn_relu(a1,P,'flow_decoder.ups.0.conv.double_conv.3','flow_decoder.ups.0.conv.double_conv.4',True,None)
xtg=ops.cfm_sample(x0.cuda(),x1.cuda(),t.cuda())[0]
ectx=engine.encoder_forward(net.encoder._blocks, xtg, torch.float32, True)
dctx=engine.decoder_forward(net.flow_decoder, ectx.feats[-1], ectx.feats[:-1][::-1], ops.time_embedding(t.cuda(),32), torch.float32, True)
for name,ref,got in (('enc L0 a1',None,None),):
    pass
def cmp(name, ref, got):
    g=got.permute(0,3,1,2).cpu()
    print(name,'err %.2e'%float((g-ref).abs().max()/ref.abs().max()),'mask mismatches',int(((g>0)!=(ref>0)).sum()), 'of', ref.numel())
cmp('dec a1',a1,dctx.layers[0][0].act); cmp('dec a2',a2,dctx.layers[0][1].act)
cmp('enc f0',sk[0],ectx.feats[0]); cmp('enc f1',b,ectx.feats[1])
e1=O.conv_bn_relu(xt,P,'encoder.inc.double_conv.0','encoder.inc.double_conv.1',True,None)
cmp('enc L0 a1',e1,ectx.layers[0][0].act)
p=O.maxpool2(sk[0]); e3=O.conv_bn_relu(p,P,'encoder.downs.0.maxpool_conv.1.double_conv.0','encoder.downs.0.maxpool_conv.1.double_conv.1',True,None)
cmp('enc L1 a1',e3,ectx.layers[1][0].act)
# pool argmax mismatches at level 0
f0=ectx.feats[0].permute(0,3,1,2).cpu()
def amax(z):
    bq,c,hh,ww=z.shape
    w=z.reshape(bq,c,hh//2,2,ww//2,2).permute(0,1,2,4,3,5).reshape(bq,c,hh//2,ww//2,4)
    return w.argmax(-1), w.max(-1).values
ia,va=amax(f0); ib,vb=amax(sk[0])
print('pool argmax mismatches', int(((ia!=ib)&(vb>0)).sum()))
