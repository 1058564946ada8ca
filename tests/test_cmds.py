"""N3: the three commands' inner functions, byte-identical to the reference's golden output files.
CPU: I/O logic with the oracle injected as the aligner.  GPU: the real path."""
import filecmp
import os

import numpy as np
import pytest

import common
import oracle
from gonomics_amd import align, cmds

MX = common.matrices()
D = common.DATA


def _cig(route):
    return [align.Cigar(r, o) for r, o in route]


def _oracle_const(a, b, sc, g):
    s, r = oracle.align_one(oracle.MODE_CONST, sc, g, 0, a, b)
    return s, _cig(r)


def _oracle_affine(a, b, sc, go, ge):
    s, r = oracle.align_one(oracle.MODE_AFFINE, sc, go, ge, a, b)
    return s, _cig(r)


def _oracle_batch(alphas, betas):
    return [_oracle_affine(a, b, MX["HumanChimpTwo"], -600, -150) for a, b in zip(alphas, betas)]


def _check_global_alignment(tmp_path, **kw):
    d = os.path.join(D, "globalAlignment")
    out = str(tmp_path / "fa_out.fa")
    text = cmds.globalAlignment(os.path.join(d, "chelsea.fa"), os.path.join(d, "eric.fa"), out, **kw)
    assert filecmp.cmp(out, os.path.join(d, "faOut_test.fa"), shallow=False)
    assert text.startswith("Alignment score is ") and "cigar is [{3 0} {3 2} {3 0}] \n" in text


def _check_cigar_to_bed(tmp_path, **kw):
    d = os.path.join(D, "cigarToBed")
    for sub, f1, f2, fi, fd, ins, dele in [
        ("sethvsraven", "seth.fa", "raven.fa", 1, 1, "affineGap_sethvsraven_ins.bed", "affineGap_sethvsraven_del.bed"),
        ("firstTest", "testRegion10kb_PanTro6.fa", "testRegion10kb_hg38.fa", 119320000, 116703287,
         "affineGap_PanTro6vshg38_ins.bed", "affineGap_PanTro6vshg38_del.bed"),
    ]:
        oi, od = str(tmp_path / "ins_tmp.bed"), str(tmp_path / "del_tmp.bed")
        cmds.GlobalAlignment_CigarToBed(os.path.join(d, sub, f1), os.path.join(d, sub, f2), "", oi, od, fi, fd, "chr1", **kw)
        assert filecmp.cmp(oi, os.path.join(d, sub, ins), shallow=False)
        assert filecmp.cmp(od, os.path.join(d, sub, dele), shallow=False)


def _check_anchor(tmp_path, **kw):
    d = os.path.join(D, "globalAlignmentAnchor")
    for idx in (1, 2):
        b1 = cmds.read_bed4(os.path.join(d, "out_hg38_gap.%d.expected.bed" % idx))
        b2 = cmds.read_bed4(os.path.join(d, "out_rheMac10_gap.%d.expected.bed" % idx))
        prefix = str(tmp_path / ("out_%d" % idx))
        cmds.gapToAlignment(b1, b2, os.path.join(d, "hg38.toy.fa"), os.path.join(d, "rheMac10.toy.fa"), "hg38", "rheMac10", prefix, **kw)
        assert filecmp.cmp(prefix + ".alignment.tsv", os.path.join(d, "out_alignment.%d.expected.tsv" % idx), shallow=False)
        assert filecmp.cmp(prefix + "_hg38_alignment.bed", os.path.join(d, "out_hg38_alignment.%d.expected.bed" % idx), shallow=False)
        assert filecmp.cmp(prefix + "_rheMac10_alignment.bed", os.path.join(d, "out_rheMac10_alignment.%d.expected.bed" % idx), shallow=False)


def test_cmd_io_logic_with_oracle(tmp_path):
    _check_global_alignment(tmp_path, const_gap=_oracle_const)
    _check_cigar_to_bed(tmp_path, affine_gap=_oracle_affine)
    _check_anchor(tmp_path, align_batch=_oracle_batch)


@pytest.mark.gpu
def test_cmds_on_gpu(gpu_lib, tmp_path):
    _check_global_alignment(tmp_path)
    _check_cigar_to_bed(tmp_path)
    _check_anchor(tmp_path)


def test_fasta_write_matches_reference_layout(tmp_path):
    """fasta.Write (line length 50) reproduces the reference-written multi-fasta fixtures byte for byte."""
    from gonomics_amd import fasta
    for nm in ("multiAlignTest.expected.fa", "multiAlignTest.expected2.fa"):
        src = os.path.join(D, "align", nm)
        out = str(tmp_path / nm)
        fasta.Write(out, fasta.Read(src))
        assert filecmp.cmp(out, src, shallow=False)
