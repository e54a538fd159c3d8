"""Host logic of the pixel-major weight-gradient route (ops.wgrad_pm_plan / applicability rules): no device needed."""
import itertools

import pytest

from pytorch_quantize_impls_amd import ops


@pytest.mark.parametrize("Cout,Cin,H,k,p,N", [(576, 192, 27, 5, 2, 256), (1152, 576, 13, 3, 1, 256), (768, 1152, 13, 3, 1, 256),
                                              (64, 64, 224, 3, 1, 32), (512, 512, 28, 3, 1, 256), (70, 40, 7, 3, 1, 3), (33, 33, 1, 3, 1, 1)])
def test_plan_invariants(Cout, Cin, H, k, p, N):
    Ho = H + 2 * p - k + 1
    ok, nslice, qa, qx = ops.wgrad_pm_plan(N, Cout, Cin, H, H, Ho, k, k, p, p, 256)
    Wq = H + 2 * p
    ktot = Ho * N * Wq
    assert ok and nslice >= 1
    assert qa % (32 * nslice) == 0 and qa >= ktot                       # whole 32-position stages per slice, every position covered
    assert qa - ktot < 32 * nslice + 32 * nslice                        # at most one stage of rounding per slice
    # the kernel reads XP rows up to q + (k - 1) * N * Wq + (k - 1) for q < qa: inside the plane
    assert qx >= qa + (k - 1) * N * Wq + (k - 1) and qx >= (H + 2 * p) * N * Wq
    # few long slices: never more workgroups than ~2 rounds of the slots unless one slice already is
    tn = 64 if k == 3 else 32
    Cpo, Cpi = -(-Cout // 64) * 64, -(-Cin // tn) * tn
    tm = 128 if (k == 3 and Cpo % 128 == 0) else 64
    tiles = (Cpo // tm) * (Cpi // tn)
    assert nslice == 1 or tiles * nslice <= 4 * 256


def test_plan_prefers_filling_the_chip_over_many_short_slices():
    # AlexNet conv3 (81 tiles of 128 x 64): 3 slices = 243 workgroups in one round beat 60 slices in 19 rounds
    ok, nslice, qa, qx = ops.wgrad_pm_plan(256, 1152, 576, 13, 13, 13, 3, 3, 1, 1, 256)
    assert nslice == 3
    # one resident slot: a single slice; plenty of slots: as many as the positions allow (<= ktot / 256)
    assert ops.wgrad_pm_plan(4, 128, 64, 9, 9, 9, 3, 3, 1, 1, 1)[1] == 1
    many = ops.wgrad_pm_plan(4, 128, 64, 9, 9, 9, 3, 3, 1, 1, 100000)[1]
    assert 1 <= many <= (9 * 4 * 11) // 256 + 1


def test_plan_respects_the_byte_budget():
    old = ops.WGRAD_GEMM_BYTES
    try:
        ops.WGRAD_GEMM_BYTES = 1 << 20
        assert ops.wgrad_pm_plan(256, 576, 192, 27, 27, 27, 5, 5, 2, 2, 256)[0] is False
        assert ops.wgrad_pm_plan(1, 64, 32, 5, 5, 5, 3, 3, 1, 1, 256)[0] is True
    finally:
        ops.WGRAD_GEMM_BYTES = old


def test_route_applicability_rules():
    x, g = (8, 64, 14, 14), (8, 96, 14, 14)
    assert ops.wgrad_pm_applicable(x, g, (3, 3), 1, 1) and ops.wgrad_pm_applicable(x, g, (5, 5), (1, 1), (1, 1))
    for bad in (((7, 7), 1, 1), ((3, 3), 2, 1), ((3, 3), 1, 2), ((1, 1), 1, 1), ((3, 5), 1, 1)):
        assert not ops.wgrad_pm_applicable(x, g, *bad)
    assert not ops.wgrad_pm_applicable((8, 16, 14, 14), g, (3, 3), 1, 1)          # too few channels for a tile
    # first-layer route: strided with few channels, or stride 1 with <= 8 channels; the s2d kernel must come out 3 x 3 or 5 x 5
    assert ops.wgrad_s2d_applicable((4, 3, 224, 224), (11, 11), 4, 1)              # AlexNet conv1: ceil(11 / 4) = 3
    assert ops.wgrad_s2d_applicable((4, 3, 32, 32), (3, 3), 1, 1)                  # VGG conv1
    assert not ops.wgrad_s2d_applicable((4, 3, 32, 32), (7, 7), 2, 1)              # ceil(7 / 2) = 4
    assert not ops.wgrad_s2d_applicable((4, 64, 32, 32), (3, 3), 1, 1)             # stride 1 with many channels: not a first layer
    assert not ops.wgrad_s2d_applicable((4, 3, 32, 32), (3, 3), 1, 2)
    assert not ops.wgrad_s2d_applicable((4, 8, 64, 64), (11, 11), 4, 1)            # 3 * 8 * 16 channels > 256


def test_plan_is_not_monotone_in_the_chunk_size_so_buffers_cover_both_plans():
    """a ragged last chunk re-plans; 64 -> 64 3x3 at 32^2 with 13 images per chunk asks for 50 slices, the
    12-image tail for 51.  ops._wgrad_pm_run sizes G3 / XP / part for the maximum of the two plans (asserted there); here:
    the counter-example exists, i.e. sizing by the full-chunk plan alone would be an out-of-bounds write."""
    full = ops.wgrad_pm_plan(13, 64, 64, 32, 32, 32, 3, 3, 1, 1, 256)
    tail = ops.wgrad_pm_plan(12, 64, 64, 32, 32, 32, 3, 3, 1, 1, 256)
    assert full[0] and tail[0]
    assert tail[1] > full[1]
    import inspect
    src = inspect.getsource(ops._wgrad_pm_run)
    assert "max(nslice, tail[1])" in src and "max(qa, tail[2])" in src and "max(qx, tail[3])" in src
