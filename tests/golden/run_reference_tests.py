"""Runs the REFERENCE's own pytest files (tests/implementations/<family>/*.py under /root/reference, in place) against the
reference package ("ref") or against this package ("ours": ``QuantTorch`` and its sub-modules resolve to
pytorch_quantize_impls_amd) and writes the per-test outcomes as JSON.  Build container only; nothing is copied.

    python tests/golden/run_reference_tests.py {ref|ours} OUT.json FILE...

Environment-only shims (both flavours alike): the two names removed from torch._jit_internal (make_golden_shim), and
``Tensor.uniform_(a, b)`` with a > b — accepted by the torch the reference was written for, rejected by torch 2.x — draws
from [b, a) instead, so that the parametrize tables at module level of the reference tests can be built."""
import importlib
import json
import os
import pkgutil
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.normpath(os.path.join(HERE, "..", "..")))

import pytest  # noqa: E402
import torch  # noqa: E402
from make_golden_shim import import_reference  # noqa: E402


REF_TESTS = "/root/reference/tests/implementations"


class _Outcomes:
    def __init__(self):
        self.results = {}

    @staticmethod
    def _key(report):
        nodeid = report.nodeid.split("implementations/")[-1]
        return nodeid if "::" in nodeid else nodeid + "::<collection>"

    def pytest_runtest_logreport(self, report):
        if report.when == "call" or (report.when == "setup" and report.outcome != "passed"):
            self.results[self._key(report)] = report.outcome

    def pytest_collectreport(self, report):
        if report.failed:
            self.results[self._key(report)] = "error"


def main():
    flavour, out, files = sys.argv[1], sys.argv[2], sys.argv[3:]
    import_reference()
    if flavour == "ours":
        for k in [k for k in sys.modules if k == "QuantTorch" or k.startswith("QuantTorch.")]:
            sys.modules.pop(k)
        pkg = importlib.import_module("pytorch_quantize_impls_amd")
        sys.modules["QuantTorch"] = pkg
        for sub in ("layers", "functions", "utils", "device"):
            m = importlib.import_module(f"pytorch_quantize_impls_amd.{sub}")
            sys.modules[f"QuantTorch.{sub}"] = m
            for mm in pkgutil.iter_modules(getattr(m, "__path__", [])):
                try:
                    sys.modules[f"QuantTorch.{sub}.{mm.name}"] = importlib.import_module(f"pytorch_quantize_impls_amd.{sub}.{mm.name}")
                except Exception:       # noqa: BLE001
                    pass
    orig = torch.Tensor.uniform_

    def uniform_(self, a=0.0, b=1.0, **kw):
        return orig(self, b, a, **kw) if a > b else orig(self, a, b, **kw)
    torch.Tensor.uniform_ = uniform_
    plug = _Outcomes()
    warnings.simplefilter("ignore")
    import random
    random.seed(0)
    torch.manual_seed(0)             # the reference's parametrize tables are random tensors drawn at import time
    pytest.main(["-q", "--continue-on-collection-errors", "-p", "no:cacheprovider", "-W", "ignore", "--rootdir", os.path.join(REF_TESTS), "-c", os.devnull, "-o", "python_files=*_test.py", *files],
                plugins=[plug])
    with open(out, "w") as fh:
        json.dump(plug.results, fh, indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
