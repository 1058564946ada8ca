"""CPU-side checks of the boundary: the C-ABI library loads, exports every symbol include/gnx_align.h
declares, and fails loudly (no CPU fallback) when there is no GPU.  No compute calls here."""
import ctypes
import os
import re

import numpy as np
import pytest

from gonomics_amd import _lib, align, dna

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "gnx_align.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(gnx_[a-z_]+)\s*\(", hdr)))


def test_header_symbols_exported():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert set(names) == set(_lib.EXPORTS)
    for nm in names:
        assert hasattr(L, nm), nm


def test_struct_layouts():
    assert ctypes.sizeof(_lib.GnxCigar) == 16 and _lib.CIGAR_DTYPE.itemsize == 16  # Go: struct{int64; uint8} on amd64
    assert _lib.GnxCigar.op.offset == 8
    assert ctypes.sizeof(_lib.GnxParams) == 8 + 25 * 8 + 4 * 8


def test_no_cpu_fallback():
    L = _lib.lib()
    if L.gnx_device_count() > 0:
        pytest.skip("a GPU is visible; covered by the gpu tests")
    with pytest.raises(_lib.GnxError) as ei:
        align.AffineGap(dna.StringToBases("ACGT"), dna.StringToBases("ACG"), align.DefaultScoreMatrix, -400, -30)
    assert ei.value.code == _lib.GNX_EDEVICE


def test_host_helpers():
    a = dna.StringToBases("ACGTNacgtn-.*")
    assert a.tolist() == list(range(13))
    assert dna.BasesToString(a) == "ACGTNacgtn-.*"
    assert dna.AllToUpper(a.copy()).tolist() == [0, 1, 2, 3, 4, 0, 1, 2, 3, 4, 10, 11, 12]
    with pytest.raises(ValueError):
        dna.StringToBases("ACGU")
    r = [align.Cigar(7, align.ColD), align.Cigar(6, align.ColM), align.Cigar(2, align.ColD)]
    assert align.PrintCigar(r) == "7D6M2D" and align.FormatCigar(r) == "[{7 2} {6 0} {2 2}]"
    assert align.View(dna.StringToBases("ACGT"), dna.StringToBases("CGT"), [align.Cigar(1, 2), align.Cigar(3, 0)]) == "ACGT\n-CGT\n"
    p = _lib.make_params(_lib.GNX_AFFINE_GAP, align.HumanChimpTwoScoreMatrix, -600, -150)
    assert p.scores[1 * 5 + 0] == -330 and p.checkersize_i == 10000
    assert np.dtype(_lib.CIGAR_DTYPE).fields["op"][1] == 8


def test_torch_restatement_of_the_synthetic_reference():
    """tests/common.py:synthetic_reference_positions_torch (used to generate the 10 M reads of config C3 on the device) equals the
    binding's numpy restatement of gnx_set_reference_synthetic, N runs and 64-bit seeds included"""
    import torch
    import common
    from gonomics_amd import _lib
    pos = np.concatenate([np.arange(0, 5000), np.arange(49999000, 50002000),
                          np.random.default_rng(1).integers(0, 4400000000, 20000)]).astype(np.int64)
    for seed in (3, 33, 0xFFFFFFFFFFFFFFF1):
        a = _lib.synthetic_reference_positions(pos, seed)
        b = common.synthetic_reference_positions_torch(torch.from_numpy(pos), seed).numpy()
        assert np.array_equal(a, b), seed
    reads, starts = common.c3_reads_torch(5, 2000, 3000000000, 3)
    assert reads.shape == (2000, 150) and reads.max() <= 4 and starts.min() >= 0
