"""pytorch_quantize_impls_amd — MI355X (gfx950) backend for the QuantTorch quantised-operator
hot path (sign / ternarize / k-bit quantize -> bit-pack -> XNOR-popcount / packed GEMM) behind
the reference's own ``functions`` / ``layers`` API.  See DESIGN.md."""
from . import functions, layers, ops, packed, utils
from .device import device
from ._lib import QtLibraryError, QtStatusError

# family aliases, as QuantTorch/{BinaryNet,TernerNet,DorefaNet,XnorNet}.py
__all__ = ["functions", "layers", "ops", "packed", "utils", "device", "QtLibraryError", "QtStatusError"]
