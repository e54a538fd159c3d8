"""A/B of two builds of libqt_hip.so on ONE box (QT_HIP_LIB selects the build): fused BinaryNet-AlexNet and fused ternary VGG-16
forwards at batch 256, plus a digest of the logits (the builds must agree bit for bit).  Usage: python tools/ab_lib.py [alexnet|vgg]"""
import hashlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench_models
from pytorch_quantize_impls_amd.layers import FusedFeatureClassifier
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "alexnet"
torch.manual_seed(0)
if which == "alexnet":
    model = bench_models.AlexNetBin(); bench_models.randomize_bn(model)
    model = model.to(dev).to(memory_format=torch.channels_last).eval()
    fused = bench_models.FusedAlexNetBin(model)
else:
    model = bench_models.TernaryVGG16(num_classes=1000, image=224); bench_models.randomize_bn(model, seed=5)
    model = model.to(dev).to(memory_format=torch.channels_last).eval()
    model.features[0].binary_input = False
    fused = FusedFeatureClassifier(model.features, model.classifier, (512, 7, 7), fold="device")
x = torch.randn((256, 3, 224, 224), device=dev).contiguous(memory_format=torch.channels_last)


def t(fn, it=60):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


with torch.no_grad():
    y = fused(x)
    dig = hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:12]
    for rep in range(3):
        print(f"{os.path.basename(os.environ.get('QT_HIP_LIB', 'libqt_hip.so')):22s} {which} fused forward {t(lambda: fused(x)):8.1f} us  logits {dig}")
