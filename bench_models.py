"""Model compositions used by bench.py / tests — the reference's topologies rebuilt from this
package's layers (the reference's own model files are not importable).

AlexNetBin: models/Alexnet/Alexnet_Bin.py:6-69 (coef = 3).
BinMLP    : benchmark/BinaryNet/MLPBin.py:6-56 cut to one hidden layer (BASELINE config C1).
"""
import torch
from torch import nn

from pytorch_quantize_impls_amd.functions import BinaryConnect
from pytorch_quantize_impls_amd.layers import BinConv2d, LinearBin


class AlexNetBin(nn.Module):
    def __init__(self, num_classes=10, coef=3):
        super().__init__()
        c = coef
        self.features = nn.Sequential(
            BinConv2d(3, 64 * c, kernel_size=11, stride=4, padding=2),
            nn.MaxPool2d(kernel_size=3, stride=2),
            nn.BatchNorm2d(64 * c), nn.Hardtanh(inplace=True), BinaryConnect(stochastic=False),

            BinConv2d(64 * c, 192 * c, kernel_size=5, padding=2),
            nn.MaxPool2d(kernel_size=3, stride=2),
            nn.BatchNorm2d(192 * c), nn.Hardtanh(inplace=True), BinaryConnect(stochastic=False),

            BinConv2d(192 * c, 384 * c, kernel_size=3, padding=1),
            nn.BatchNorm2d(384 * c), nn.Hardtanh(inplace=True), BinaryConnect(stochastic=False),

            BinConv2d(384 * c, 256 * c, kernel_size=3, padding=1),
            nn.BatchNorm2d(256 * c), nn.Hardtanh(inplace=True), BinaryConnect(stochastic=False),

            BinConv2d(256 * c, 256, kernel_size=3, padding=1),
            nn.MaxPool2d(kernel_size=3, stride=2),
            nn.BatchNorm2d(256), nn.Hardtanh(inplace=True),
        )
        self.classifieur = nn.Sequential(
            BinaryConnect(stochastic=False), LinearBin(256 * 6 * 6, 4096),
            nn.BatchNorm1d(4096), nn.Hardtanh(inplace=True),
            BinaryConnect(stochastic=False), LinearBin(4096, 4096),
            nn.BatchNorm1d(4096), nn.Hardtanh(inplace=True),
            BinaryConnect(stochastic=False), LinearBin(4096, num_classes),
            nn.LogSoftmax(dim=1),
        )

        # the first layer sees real pixels: tell it so (binary_input hint of the layers), which skips the
        # +-1 detection of un-tagged inputs and its host sync
        self.features[0].binary_input = False

    def clip(self):
        for layer in self.modules():
            if isinstance(layer, (BinConv2d, LinearBin)):
                layer.clamp()

    def forward(self, x):
        x = self.features(x)
        x = x.reshape(x.size(0), 256 * 6 * 6)   # logical NCHW order whatever the memory format
        return self.classifieur(x)


class AlexNetFloat(nn.Module):
    """The AlexNetBin topology (models/Alexnet/Alexnet_Bin.py:12-54) written with plain nn.Conv2d / nn.Linear: the float network
    the reference's converters take (utils/convertor.py:40-58 replace exactly these two classes)."""

    def __init__(self, num_classes=10, coef=3):
        super().__init__()
        c = coef
        self.features = nn.Sequential(
            nn.Conv2d(3, 64 * c, kernel_size=11, stride=4, padding=2),
            nn.MaxPool2d(kernel_size=3, stride=2),
            nn.BatchNorm2d(64 * c), nn.Hardtanh(inplace=True), BinaryConnect(stochastic=False),

            nn.Conv2d(64 * c, 192 * c, kernel_size=5, padding=2),
            nn.MaxPool2d(kernel_size=3, stride=2),
            nn.BatchNorm2d(192 * c), nn.Hardtanh(inplace=True), BinaryConnect(stochastic=False),

            nn.Conv2d(192 * c, 384 * c, kernel_size=3, padding=1),
            nn.BatchNorm2d(384 * c), nn.Hardtanh(inplace=True), BinaryConnect(stochastic=False),

            nn.Conv2d(384 * c, 256 * c, kernel_size=3, padding=1),
            nn.BatchNorm2d(256 * c), nn.Hardtanh(inplace=True), BinaryConnect(stochastic=False),

            nn.Conv2d(256 * c, 256, kernel_size=3, padding=1),
            nn.MaxPool2d(kernel_size=3, stride=2),
            nn.BatchNorm2d(256), nn.Hardtanh(inplace=True),
        )
        self.classifieur = nn.Sequential(
            BinaryConnect(stochastic=False), nn.Linear(256 * 6 * 6, 4096),
            nn.BatchNorm1d(4096), nn.Hardtanh(inplace=True),
            BinaryConnect(stochastic=False), nn.Linear(4096, 4096),
            nn.BatchNorm1d(4096), nn.Hardtanh(inplace=True),
            BinaryConnect(stochastic=False), nn.Linear(4096, num_classes),
            nn.LogSoftmax(dim=1),
        )

    def forward(self, x):
        x = self.features(x)
        x = x.reshape(x.size(0), 256 * 6 * 6)
        return self.classifieur(x)


def alexnet_xnor(num_classes=10, coef=3):
    """BASELINE config 3 in its XNOR-Net flavour: ``xnor_net_convert`` of the float topology — XNORConv2d(dim=[0, 1])
    / LinearXNOR everywhere (utils/convertor.py:54-58; fresh layers, weights are not copied, like upstream).  The first layer
    sees real pixels: its ``binary_input`` hint is set like AlexNetBin's."""
    from pytorch_quantize_impls_amd.utils import xnor_net_convert
    net = xnor_net_convert(AlexNetFloat(num_classes, coef))
    net.features[0].binary_input = False
    return net


class FusedAlexNetBin(nn.Module):
    """Inference form of an (eval-mode) AlexNetBin (layers.FusedFeatureClassifier): every BinConv2d emits
    BatchNorm-threshold bits, MaxPool runs on bits, activations between binarised layers exist only as bit planes.
    Shares the parameters of the model it was built from.  ``fuse_conv=False`` keeps fp32 conv outputs and fuses
    only the [MaxPool, BatchNorm, Hardtanh, BinaryConnect] runs (one kernel each)."""

    def __init__(self, model: AlexNetBin, fuse_conv: bool = True, fold=None):
        super().__init__()
        from pytorch_quantize_impls_amd.layers import FusedFeatureClassifier
        assert not model.training, "fuse an eval-mode model"
        self.net = FusedFeatureClassifier(model.features, model.classifieur, (256, 6, 6), fuse_conv=fuse_conv, fold=fold)

    def forward(self, x):
        return self.net(x)


class BinMLP(nn.Module):
    def __init__(self, in_features=784, hidden=512, out_features=10):
        super().__init__()
        self.linear1 = LinearBin(in_features, hidden)
        self.norm1 = nn.BatchNorm1d(hidden, eps=1e-4, momentum=0.15)
        self.act = BinaryConnect()
        self.linear2 = LinearBin(hidden, out_features)

    def forward(self, x):
        x = torch.relu(self.linear1(x.view(x.shape[0], -1)))
        x = self.act(self.norm1(x))
        return torch.log_softmax(self.linear2(x), dim=1)


def randomize_bn(model, seed=0):
    """Give BatchNorm layers non-trivial eval statistics so sign(BN(x)) is a real threshold."""
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 3)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) * 50 + 50)
            m.weight.data.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(m.num_features, generator=g) * 0.1)


# ---- BASELINE configs C4 / C5 as parity-test models -----------------------------

class _DorefaBlock(nn.Module):
    """Residual block of the C4 net: two 3x3 DorefaConv2d (1-bit weights) + BatchNorm, k-bit activation
    quantiser applied to the UNCLAMPED relu(bn(.)) (models/samples/ResNet_Dorefa.py pattern), 1x1 stride-2
    DorefaConv2d shortcut when the shape changes; the second conv consumes the first one's output (the
    upstream block feeds x twice, models/Resnet/Resnet_bin.py:29 — fixed here as SURVEY 8d specifies)."""

    def __init__(self, cin, cout, stride, w_bits, a_bits):
        super().__init__()
        from pytorch_quantize_impls_amd.layers import DorefaConv2d
        from pytorch_quantize_impls_amd.functions import nnDorefaQuant
        self.conv1 = DorefaConv2d(cin, cout, 3, stride=stride, padding=1, bias=False, bit_width=w_bits)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = DorefaConv2d(cout, cout, 3, stride=1, padding=1, bias=False, bit_width=w_bits)
        self.bn2 = nn.BatchNorm2d(cout)
        self.quant = nnDorefaQuant(a_bits)
        self.shortcut = None
        if stride != 1 or cin != cout:
            self.shortcut = nn.Sequential(DorefaConv2d(cin, cout, 1, stride=stride, bias=False, bit_width=w_bits),
                                          nn.BatchNorm2d(cout))

    def forward(self, x):
        out = self.quant(torch.relu(self.bn1(self.conv1(x))))
        out = self.bn2(self.conv2(out))
        out = out + (x if self.shortcut is None else self.shortcut(x))
        return self.quant(torch.relu(out))


class DorefaResNet18(nn.Module):
    """C4: DoReFa ResNet-18 W1A4 for 3x32x32 inputs (fp32 stem, avg_pool 4, Linear(512))."""

    def __init__(self, num_classes=10, w_bits=1, a_bits=4):
        super().__init__()
        from pytorch_quantize_impls_amd.functions import nnDorefaQuant
        self.stem = nn.Conv2d(3, 64, 3, padding=1, bias=False)
        self.bn = nn.BatchNorm2d(64)
        self.quant = nnDorefaQuant(a_bits)
        blocks, cin = [], 64
        for cout, stride in ((64, 1), (64, 1), (128, 2), (128, 1), (256, 2), (256, 1), (512, 2), (512, 1)):
            blocks.append(_DorefaBlock(cin, cout, stride, w_bits, a_bits))
            cin = cout
        self.blocks = nn.Sequential(*blocks)
        self.linear = nn.Linear(512, num_classes)

    def forward(self, x):
        out = self.quant(torch.relu(self.bn(self.stem(x))))
        out = self.blocks(out)
        out = torch.nn.functional.avg_pool2d(out, 4)
        return self.linear(out.reshape(out.size(0), -1))


class _FusedDorefaBlock(nn.Module):
    """Inference form of _DorefaBlock: activations between the DorefaConv2d layers exist only as int8 code planes
    (layers.FusedBnDorefaQuant folds BatchNorm + shortcut add + ReLU + quantiser into one pass per conv output)."""

    def __init__(self, blk: _DorefaBlock, a_bits: int, fuse_conv: bool = True, halo: int = 1, fold=None):
        super().__init__()
        from pytorch_quantize_impls_amd.layers import FusedBnDorefaQuant, FusedDorefaConvBnQuant
        self.fuse_conv = fuse_conv
        if fuse_conv:       # the whole tail in the conv epilogue: no fp32 conv output at all
            # every consumer is a 3x3 conv with padding 1 (or the 1x1 shortcut): planes carry a 1-pixel zero halo
            self.c1 = FusedDorefaConvBnQuant(blk.conv1, blk.bn1, a_bits, out_halo=halo, fold=fold)
            self.c2 = FusedDorefaConvBnQuant(blk.conv2, blk.bn2, a_bits, out_halo=halo, fold=fold)
        else:               # fp32 conv output + one fused elementwise pass
            self.conv1, self.conv2 = blk.conv1, blk.conv2
            self.q1 = FusedBnDorefaQuant(blk.bn1, a_bits, fold=fold)
            self.q2 = FusedBnDorefaQuant(blk.bn2, a_bits, fold=fold)
        self.sc_conv, self.sc_bn = (blk.shortcut[0], blk.shortcut[1]) if blk.shortcut is not None else (None, None)

    def forward(self, act):
        if self.fuse_conv:
            if self.sc_conv is None:
                return self.c2(self.c1(act), residual=act)
            return self.c2(self.c1(act), residual_conv=(self.sc_conv, act), residual_bn=self.sc_bn)   # shortcut conv + BN: one launch
        res = act if self.sc_conv is None else self.sc_conv(act)
        return self.q2(self.conv2(self.q1(self.conv1(act))), residual=res, residual_bn=self.sc_bn)


class FusedDorefaResNet18(nn.Module):
    """Inference form of an eval-mode DorefaResNet18 (shares its parameters)."""

    def __init__(self, model: DorefaResNet18, a_bits: int = 4, fuse_conv: bool = True, halo: int = 1, fold=None):
        super().__init__()
        from pytorch_quantize_impls_amd.layers import FusedBnDorefaQuant
        assert not model.training, "fuse an eval-mode model"
        self.stem, self.linear = model.stem, model.linear
        self.q0 = FusedBnDorefaQuant(model.bn, a_bits, out_halo=halo if fuse_conv else 0, fold=fold)
        self.blocks = nn.Sequential(*[_FusedDorefaBlock(b, a_bits, fuse_conv, halo, fold) for b in model.blocks])

    def forward(self, x):
        out = self.blocks(self.q0(self.stem(x))).avg_pool2d(4)      # codes -> fp32 image -> avg_pool2d in one pass
        return self.linear(out.reshape(out.size(0), -1))


class TernaryVGG16(nn.Module):
    """C5: VGG-16 (13 conv3x3 p1 + 3 FC) from TerConv2d / LinearTer with BinaryConnect activations, in the
    conv -> BatchNorm -> Hardtanh -> BinaryConnect pattern of models/FullNet/terMNIST.py:48-59.  ``image`` sets the
    input resolution (224 in the config; the classifier width follows)."""
    CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M")

    def __init__(self, num_classes=1000, image=224, fc=4096):
        super().__init__()
        from pytorch_quantize_impls_amd.layers import LinearTer, TerConv2d
        layers, cin = [], 3
        for v in self.CFG:
            if v == "M":
                layers.append(nn.MaxPool2d(2, 2))
                continue
            layers += [TerConv2d(cin, v, 3, padding=1), nn.BatchNorm2d(v), nn.Hardtanh(), BinaryConnect(stochastic=False)]
            cin = v
        self.features = nn.Sequential(*layers)
        side = image // 32
        self.classifier = nn.Sequential(LinearTer(512 * side * side, fc), nn.BatchNorm1d(fc), nn.Hardtanh(),
                                        BinaryConnect(stochastic=False), LinearTer(fc, fc), nn.BatchNorm1d(fc),
                                        nn.Hardtanh(), BinaryConnect(stochastic=False), LinearTer(fc, num_classes))

    def forward(self, x):
        x = self.features(x)
        return self.classifier(x.reshape(x.size(0), -1))


class TrainFusedAlexNetBin(nn.Module):
    """AlexNetBin for TRAINING with every [MaxPool2d?, BatchNorm, Hardtanh, BinaryConnect] run as one FusedTrainPoolBnSign
    (layers.fuse_sequential_training): shares all parameters / buffers with ``model``.  The BinaryConnect that opens the
    classifier is applied before the flatten (sign commutes with a reshape), so the last feature block is fused too."""

    def __init__(self, model: AlexNetBin):
        super().__init__()
        from pytorch_quantize_impls_amd.layers import fuse_sequential_training
        f, c = list(model.features.children()), list(model.classifieur.children())
        self.features = fuse_sequential_training(nn.Sequential(*f, c[0]))
        self.classifieur = fuse_sequential_training(nn.Sequential(*c[1:]))

    def forward(self, x):
        x = self.features(x)
        return self.classifieur(x.reshape(x.size(0), 256 * 6 * 6))


class _TrainFusedDorefaBlock(nn.Module):
    """_DorefaBlock for TRAINING with BatchNorm (+ shortcut add + ReLU + quantiser) behind every conv as one
    layers.FusedTrainBnActQuant node (shares the block's convs and BatchNorm modules)."""

    def __init__(self, blk: _DorefaBlock, a_bits: int):
        super().__init__()
        from pytorch_quantize_impls_amd.layers import FusedTrainBnActQuant
        self.conv1, self.conv2 = blk.conv1, blk.conv2
        self.q1 = FusedTrainBnActQuant(blk.bn1, a_bits, relu=True)
        self.q2 = FusedTrainBnActQuant(blk.bn2, a_bits, relu=True)
        self.sc_conv = blk.shortcut[0] if blk.shortcut is not None else None
        self.sc_bn = FusedTrainBnActQuant(blk.shortcut[1]) if blk.shortcut is not None else None

    def forward(self, x):
        res = x if self.sc_conv is None else self.sc_bn(self.sc_conv(x))
        return self.q2(self.conv2(self.q1(self.conv1(x))), residual=res)


class TrainFusedDorefaResNet18(nn.Module):
    """DorefaResNet18 for TRAINING on the fused BatchNorm chains (shares all parameters / buffers with ``model``)."""

    def __init__(self, model: DorefaResNet18, a_bits: int = 4):
        super().__init__()
        from pytorch_quantize_impls_amd.layers import FusedTrainBnActQuant
        self.stem, self.linear = model.stem, model.linear
        self.q0 = FusedTrainBnActQuant(model.bn, a_bits, relu=True)
        self.blocks = nn.Sequential(*[_TrainFusedDorefaBlock(b, a_bits) for b in model.blocks])

    def forward(self, x):
        out = self.blocks(self.q0(self.stem(x)))
        out = torch.nn.functional.avg_pool2d(out, 4)
        return self.linear(out.reshape(out.size(0), -1))
