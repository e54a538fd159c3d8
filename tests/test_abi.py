"""The C-ABI library loads and exports every symbol include/qt_hip.h declares (no GPU needed:
only argument validation paths are exercised, nothing is launched)."""
import ctypes

import pytest

from pytorch_quantize_impls_amd import _lib


@pytest.fixture(scope="module")
def lib():
    if not _lib.is_built():
        import __graft_entry__ as g
        g.build()
    return _lib.load()


def test_header_symbols_all_bound_and_exported(lib):
    declared = _lib.header_declared_functions()
    assert declared, "header parse found nothing"
    assert sorted(_lib.SIGNATURES) == declared, "python SIGNATURES out of sync with include/qt_hip.h"
    for name in declared:
        assert hasattr(lib, name), f"{name} not exported by libqt_hip.so"


def test_identification(lib):
    assert _lib.version() >= 100
    assert _lib.target_arch() == "gfx950"
    assert _lib.strerror(0) == "ok"
    for code in (-1, -2, -3, -4, -5, -99):
        assert isinstance(_lib.strerror(code), str) and _lib.strerror(code)


def test_argument_validation_returns_status_not_crash(lib):
    null = ctypes.c_void_p(0)
    i64 = ctypes.c_int64
    assert lib.qt_binarize_f32(null, null, i64(-1), null) == -1
    assert lib.qt_binarize_f32(null, null, i64(4), null) == -1
    assert lib.qt_binarize_f32(null, null, i64(0), null) == 0           # empty input is fine
    assert lib.qt_dorefa_quantize_f32(null, null, i64(0), 0, null) == -1  # bit_width out of range
    fake = ctypes.c_void_p(0x1000)
    # packed row stride must be a multiple of 4 words and cover ceil(K/32)
    assert lib.qt_sign_pack_f32(fake, i64(64), fake, i64(3), null, i64(0), i64(1), i64(64), null) == -2
    assert lib.qt_sign_pack_f32(fake, i64(64), fake, i64(0), null, i64(0), i64(1), i64(64), null) == -2
    assert lib.qt_sign_pack_f32(fake, i64(8), fake, i64(4), null, i64(0), i64(1), i64(64), null) == -1
    assert lib.qt_xnor_gemm(fake, i64(4), fake, i64(6), null, fake, i64(8), i64(8), i64(8), i64(64), null) == -2
    assert lib.qt_xnor_gemm(fake, i64(4), fake, i64(4), null, fake, i64(4), i64(8), i64(8), i64(64), null) == -1
    assert lib.qt_xnor_gemm(fake, i64(4), fake, i64(4), null, fake, i64(8), i64(0), i64(8), i64(64), null) == 0
    assert lib.qt_tern_gemm(fake, i64(4), fake, null, i64(4), null, fake, i64(8), i64(8), i64(8), i64(64), null) == -1


def test_call_wrapper_raises():
    with pytest.raises(_lib.QtStatusError):
        _lib.call("qt_binarize_f32", ctypes.c_void_p(0), ctypes.c_void_p(0), ctypes.c_int64(-5),
                  ctypes.c_void_p(0))


def test_cpu_tensor_is_rejected_loudly():
    import torch
    from pytorch_quantize_impls_amd import ops
    with pytest.raises(TypeError):
        ops.binarize(torch.zeros(4))
    with pytest.raises(TypeError):
        ops.sign_pack(torch.zeros(2, 32))


def test_the_product_package_never_touches_the_oracle():
    """oracle/ is test infrastructure (its header says so): no module of the shipped package imports, loads or executes anything
    from it — checked on the sources and on sys.modules after importing every sub-package."""
    import importlib
    import pathlib
    import re
    import sys
    root = pathlib.Path(__file__).resolve().parents[1] / "pytorch_quantize_impls_amd"
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|[\"']oracle[\"'/\\]|_ref[/\\]", re.M)      # imports and path literals
    for f in root.rglob("*.py"):
        assert not pat.search(f.read_text()), f
    before = {m for m in sys.modules if m == "oracle" or m.startswith("oracle.")}
    for name in ("pytorch_quantize_impls_amd", "pytorch_quantize_impls_amd.functions", "pytorch_quantize_impls_amd.layers",
                 "pytorch_quantize_impls_amd.utils", "pytorch_quantize_impls_amd.lazy", "pytorch_quantize_impls_amd.ops"):
        importlib.import_module(name)
    after = {m for m in sys.modules if m == "oracle" or m.startswith("oracle.")}
    assert after == before
