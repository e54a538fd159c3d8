import copy, sys, torch
sys.path.insert(0, ".")
import bench_models
from pytorch_quantize_impls_amd import lazy_train
dev = torch.device("cuda:0")
torch.manual_seed(22)
m = bench_models.DorefaResNet18(w_bits=1, a_bits=4)
bench_models.randomize_bn(m, seed=3)
m = m.to(dev).to(memory_format=torch.channels_last).train()
x = torch.randn(32, 3, 32, 32, device=dev).contiguous(memory_format=torch.channels_last)
t = torch.randint(0, 10, (32,), device=dev)

class Stem(torch.nn.Module):
    def __init__(s, bn, q):
        super().__init__(); s.bn, s.quant = bn, q
    def forward(s, x): return s.quant(torch.relu(s.bn(x)))

def run(kind):
    tw = copy.deepcopy(m)
    if kind == "explicit":
        net = bench_models.TrainFusedDorefaResNet18(tw); net.q0 = Stem(tw.bn, tw.quant)
    else:
        net = tw
    loss = torch.nn.functional.cross_entropy(net(x), t); loss.backward(); torch.cuda.synchronize()
    return loss.detach(), {k: p.grad.clone() for k, p in tw.named_parameters()}
la, ga = run("explicit"); lb, gb = run("explicit"); lc, gc = run("graph"); ld, gd = run("graph")
print("loss", float(la), float(lb), float(lc))
for name, g1, g2 in (("explicit vs explicit", ga, gb), ("graph vs graph", gc, gd), ("explicit vs graph", ga, gc)):
    bad = [(k, float((g1[k] - g2[k]).abs().max() / (g1[k].abs().max() + 1e-30))) for k in g1 if not torch.equal(g1[k], g2[k])]
    print(name, len(bad), bad[:6])
