import sys; import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from pytorch_quantize_impls_amd import ops
dev=torch.device('cuda:0')
torch.manual_seed(0)
for (N,C,H,W,Cout,k,s,p) in [(4,3,224,224,192,11,4,2),(3,3,37,45,64,11,4,2),(2,3,32,32,40,7,2,3),(2,1,28,28,33,5,2,2),(5,3,64,64,96,9,3,0), (256,3,224,224,192,11,4,2)]:
    x=torch.randn(N,C,H,W,device=dev).contiguous(memory_format=torch.channels_last)
    Ho=(H+2*p-k)//s+1; Wo=(W+2*p-k)//s+1
    g=torch.randn(N,Cout,Ho,Wo,device=dev).contiguous(memory_format=torch.channels_last)
    got=ops.conv2d_grad_weight_s2d(x,g,(Cout,C,k,k),s,p)
    if got is None: print('none',N,C,H,W,Cout,k,s,p); continue
    if N<=8:
        ref=torch.nn.grad.conv2d_weight(x.double(),(Cout,C,k,k),g.double(),stride=s,padding=p)
    else:
        ref=torch.nn.grad.conv2d_weight(x,(Cout,C,k,k),g,stride=s,padding=p).double()
    print((N,C,H,W,Cout,k,s,p), 'err', float((got.double()-ref).abs().max()/ref.abs().max()))
    if N>8:
        import time
        for fn,name in ((lambda: ops.conv2d_grad_weight_s2d(x,g,(Cout,C,k,k),s,p),'s2d pm'),(lambda: torch.nn.grad.conv2d_weight(x,(Cout,C,k,k),g,stride=s,padding=p),'miopen')):
            fn(); torch.cuda.synchronize(); t0=time.perf_counter()
            for _ in range(5): fn()
            torch.cuda.synchronize(); print(name, (time.perf_counter()-t0)/5*1e3,'ms')
