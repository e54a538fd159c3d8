"""How well do the fused-chain gradients of DorefaResNet18 agree with the module graph's — and how well does the module graph agree with
ITSELF when the input moves by one part in 10^6 (the noise floor of a quantised net: codes next to a rounding boundary flip)?"""
import copy, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench_models
dev = torch.device("cuda:0")
torch.manual_seed(4)
m = bench_models.DorefaResNet18(w_bits=int(os.environ.get("W_BITS", "1")), a_bits=4)
bench_models.randomize_bn(m, seed=3)
m = m.to(dev).to(memory_format=torch.channels_last).train()
x = torch.randn(64, 3, 32, 32, device=dev).contiguous(memory_format=torch.channels_last)
t = torch.randint(0, 10, (64,), device=dev)
def grads(net, model, inp):
    model.zero_grad(set_to_none=True)
    loss = torch.nn.functional.cross_entropy(net(inp), t); loss.backward()
    return float(loss), {k: p.grad.double().flatten().clone() for k, p in model.named_parameters()}
m_a, m_b, m_c = copy.deepcopy(m), copy.deepcopy(m), copy.deepcopy(m)
l0, g0 = grads(m_a, m_a, x)
l1, g1 = grads(m_b, m_b, x * (1 + 1e-6))
l2, g2 = grads(bench_models.TrainFusedDorefaResNet18(m_c), m_c, x)
cos = lambda a, b: float((a * b).sum() / (a.norm() * b.norm() + 1e-300))
print("loss module", l0, "perturbed", l1, "fused", l2)
for k in g0:
    print(f"{k:40s} module-vs-perturbed {cos(g0[k], g1[k]):.4f}   fused-vs-module {cos(g0[k], g2[k]):.4f}")
