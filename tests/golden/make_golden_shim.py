"""Environment-only import shim for the golden-vector generators (build container only): two names
removed from torch._jit_internal are re-added as identities and QuantTorch/__init__.py (which pulls optuna / torchvision /
progress, absent here) is bypassed by pre-registering an empty package object.  Nothing of the reference is copied."""
import os
import sys
import types
import warnings

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)

from pytorch_quantize_impls_amd import synth  # noqa: E402,F401


def import_reference():
    import torch._jit_internal as J
    for n in ("weak_module", "weak_script_method"):
        if not hasattr(J, n):
            setattr(J, n, lambda x: x)
    pkg = types.ModuleType("QuantTorch")
    pkg.__path__ = [os.path.join(REF, "QuantTorch")]
    sys.modules["QuantTorch"] = pkg
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import QuantTorch.functions as RF
        import QuantTorch.layers as RL
    return RF, RL
