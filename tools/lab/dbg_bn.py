import sys; sys.path.insert(0,'.')
import torch
from geot_amd import _lib
from geot_amd.ext._common import call, ptr
dev='cuda:0'
for (b,c,l) in [(3,8,4099),(2,5,4097),(1,4,4100)]:
    x=torch.randn(b,c,l,device=dev)*2+0.5
    S=int(_lib.load().geot_bn_slices(b,c,l))
    part=torch.empty(b,c,S,2,device=dev)
    call("geot_bn_stats", x.device, b,c,l, ptr(x), ptr(part))
    got=part.sum((0,2),dtype=torch.float64)
    want=torch.stack([x.double().sum((0,2)), x.double().square().sum((0,2))],1)
    print(b,c,l,S,(got-want).abs().max().item(), want.abs().max().item())
    per_row=part[:,:,0,0].double(); wr=x.double().sum(2)
    print((per_row-wr).abs().flatten().tolist())
