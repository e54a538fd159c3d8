import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_quantize_impls_amd import ops
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "conv2"
N, Cin, Cout, H, k, p = {"conv2": (256, 192, 576, 27, 5, 2), "vgg56": (256, 256, 256, 56, 3, 1), "vgg224": (32, 64, 64, 224, 3, 1),
                         "conv3": (256, 576, 1152, 13, 3, 1), "vgg28": (256, 512, 512, 28, 3, 1)}[name]
x = (torch.randint(0, 2, (N, Cin, H, H), device=dev).float() * 2 - 1).contiguous(memory_format=torch.channels_last)
Ho = H + 2 * p - k + 1
go = torch.randn((N, Cout, Ho, Ho), device=dev).contiguous(memory_format=torch.channels_last)
ops.WGRAD_WORKGROUPS = int(os.environ.get('WG', '1024'))
ops.WGRAD_PM_WORKGROUPS = int(os.environ.get('WGPM', str(ops.WGRAD_PM_WORKGROUPS)))
fn = ops.conv2d_grad_weight_pm if os.environ.get('ROUTE', 'gemm') == 'pm' else ops.conv2d_grad_weight_gemm
for _ in range(int(os.environ.get('ITERS', '12'))):
    fn(x, go, (k, k), p)
torch.cuda.synchronize()
