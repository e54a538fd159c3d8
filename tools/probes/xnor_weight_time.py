"""qt_xnor_weight_f32 (alpha = column means of |W|, sign(W) * alpha): time and error against fp64 at the XNOR-AlexNet weight shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, time
from pytorch_quantize_impls_amd import ops
dev=torch.device("cuda:0")
for shape,lead in (((576,192,5,5),2),((1152,576,3,3),2),((4096,9216),1),((3000,100),1),((2048,4100),1)):
    w=torch.randn(*shape,device=dev)*0.1
    wq,a=ops.xnor_weight(w,lead)
    ref=w.double().abs().mean(tuple(range(lead)),keepdim=True)
    err=float(((a.double()-ref).abs().max()/ref.abs().max()))
    refq=(torch.sign(w.double())*ref)
    errq=float((wq.double()-refq).abs().max()/refq.abs().max())
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(10): ops.xnor_weight(w,lead)
    torch.cuda.synchronize(); print(shape, f"{(time.perf_counter()-t0)/10*1e6:.0f} us", err, errq)
