"""Public layer surface — the names of QuantTorch/layers/__init__.py for the four hot-path
families, plus the Lin / Log fixed-point layers."""
from .binary_layers import LinearBin, BinConv2d, ShiftNormBatch1d, ShiftNormBatch2d
from .dorefa_layers import LinearDorefa, DorefaConv2d
from .terner_layers import LinearTer, TerConv2d
from .xnor_layers import LinearXNOR, XNORConv2d
from .log_lin_layers import LinearQuant, QuantConv2d
from .common import QLayer
from .fused import (FusedTrainPoolBnSign, FusedTrainBnActQuant, fuse_sequential_training, CodeMaxPool, FusedBnDorefaQuant, FusedDorefaConvBnQuant, FusedPoolBnSign, FusedConvPoolBnSign, FusedFeatureClassifier, PackedMaxPool, fuse_sequential, fold_batchnorm,
                    permute_fc_weight_hwc)
