"""pytorch_quantize_impls_amd — MI355X (gfx950) backend for the QuantTorch quantised-operator
hot path (sign / ternarize / k-bit quantize -> bit-pack -> XNOR-popcount / packed GEMM) behind
the reference's own ``functions`` / ``layers`` API.  See DESIGN.md."""
from . import functions, layers, ops, packed, utils
from .device import device
from ._lib import QtLibraryError, QtStatusError

# family aliases, as QuantTorch/{BinaryNet,TernerNet,DorefaNet,XnorNet,LogLinNet}.py (imported lazily: `import pkg.BinaryNet`)
__all__ = ["functions", "layers", "ops", "packed", "utils", "device", "QtLibraryError", "QtStatusError",
           "BinaryNet", "TernerNet", "DorefaNet", "XnorNet", "LogLinNet"]


def __getattr__(name):
    if name in ("BinaryNet", "TernerNet", "DorefaNet", "XnorNet", "LogLinNet"):
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
