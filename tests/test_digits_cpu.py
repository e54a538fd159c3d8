"""Host logic of the integer form of LinearXNOR (round 4): the fixed-point digit table of alpha and the split-K plan.  CPU only."""
import numpy as np
import pytest
import torch

from pytorch_quantize_impls_amd import ops


@pytest.mark.parametrize("M,N,ld", [(768, 4096, 9216), (768, 4096, 4096), (768, 10, 4096), (768, 1000, 4096), (3, 7, 128), (768, 4096, 128),
                                    (12288, 4096, 9216), (96, 33, 1024)])
def test_splitk_plan_divides_the_padded_row_into_whole_stages(M, N, ld):
    kslice, nslice = ops.splitk_plan(M, N, ld)
    assert kslice * nslice == ld and kslice % 64 == 0 and nslice >= 1
    tiles = ((M + 255) // 256) * ((N + ops._pick_tile_n(N) - 1) // ops._pick_tile_n(N))
    assert nslice == 1 or (tiles * nslice <= 256 and kslice >= 512)
    # the largest admissible slice count was taken
    for d in range(nslice + 1, ld // 64 + 1):
        if (ld // 64) % d == 0:
            assert tiles * d > 256 or (ld // 64 // d) * 64 < 512


def test_pick_tile_n_follows_the_kernel_header():
    # csrc/mfma_gemm_kernel.h pick_tile_n: fewest padded columns, ties to the wider tile
    assert [ops._pick_tile_n(n) for n in (10, 64, 65, 128, 192, 256, 576, 1000, 1152, 4096)] == [64, 64, 128, 128, 192, 256, 192, 256, 192, 256]


@pytest.mark.parametrize("scale", [1e-3, 0.04, 1.0, 300.0])
def test_alpha_digits_reconstruct_alpha_to_2_pow_minus_21_of_its_maximum(scale):
    torch.manual_seed(int(scale * 1000) % 97)
    a = torch.rand(1000) * scale
    a[3] = 0.0
    dg = ops.alpha_digits(a)
    assert dg is not None and dg.K == 1000 and dg.table.numel() == ops.code_ld_bytes(1000)
    t = dg.table[:1000].to(torch.int64)
    d0, d1, d2 = t & 0xff, (t >> 8) & 0xff, (t >> 16) & 0xff
    assert int(d0.max()) <= 127 and int(d1.max()) <= 127 and int(d2.max()) <= 127
    assert int(dg.table[1000:].abs().max()) == 0                      # zero past K: the planes' pad columns
    A = (d0 << 14) | (d1 << 7) | d2
    s = float(dg.scale)
    assert np.log2(s) == round(np.log2(s))                               # a power of two
    rec = A.double() * s
    assert float((rec - a.double()).abs().max()) <= 2.0 ** -21 * float(a.max())
    assert int(A.max()) < (1 << 21) and int(A[3]) == 0


def test_alpha_digits_decline_non_finite_or_negative_scales():
    for bad in (float("inf"), float("nan"), -1.0):
        a = torch.rand(64)
        a[5] = bad
        assert ops.alpha_digits(a) is None
    z = ops.alpha_digits(torch.zeros(40))                               # all-zero alpha: every digit 0, scale 1
    assert z is not None and int(z.table.abs().max()) == 0 and float(z.scale) == 1.0
