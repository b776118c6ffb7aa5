import sys; sys.path.insert(0,'/root/repo')
import torch, vp_suite_amd as v
L=v._lib.lib()
torch.manual_seed(0)
x=v.ops.to_channels_last(torch.rand(32,4,64,32,32,device='cuda')); W=torch.randn(384,160,3,3,device='cuda')*0.03; b=torch.zeros(384,device='cuda'); pw=[torch.randn(1,96,32,32,device='cuda')*0.1 for _ in range(3)]
outs=[]
with torch.no_grad():
    for e in (0,1,2,3):
        L.vpx_set_option(v._lib.OPT_EXPERIMENT, e)
        o,_,c=v.ops.convlstm_seq(x,None,None,W,b,*pw,seq_len=4,in_channels=64,precision='bf16x3')
        outs.append((o.clone(),c.clone()))
L.vpx_set_option(v._lib.OPT_EXPERIMENT, 0)
for e in (1,2,3): print(e, torch.equal(outs[0][0],outs[e][0]), torch.equal(outs[0][1],outs[e][1]))
