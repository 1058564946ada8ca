#!/usr/bin/env python3
"""Regenerate tests/golden/ from the reference checkout (run in the build container only).

The GPU box has no /root/reference, so everything the tests need is committed here as *data*:
  * align_tables.json  -- the input/expected-output tables of the reference's table-driven tests
      align/affineGap_test.go:11-36   (affineAlignTests, affineAlignChunkTests)
      align/affineGap_test.go:120-192 (AffineGapLocal / GoAffineGapLocalEngine known answers)
      align/view_test.go:9-26         (alignTests, ConstGap)
      cmd/globalAlignment/globalAlignment_test.go:13-14 (toad / ahsoka)
    extracted by regex from the Go test sources (only the string/int literals are kept).
  * data/...           -- fixture *data files* the reference's tests read / diff against
      cmd/globalAlignmentAnchor/testdata/{hg38.toy.fa,rheMac10.toy.fa,out_alignment.{1,2}.expected.tsv,
                                          out_{hg38,rheMac10}_gap.{1,2}.expected.bed}
      cmd/cigarToBed/testdata/{firstTest,sethvsraven}/*
      cmd/globalAlignment/testdata/*.fa
No reference source text is copied.
"""
import json
import os
import re
import shutil
import sys

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def go_unquote(s):
    return s.encode("utf-8").decode("unicode_escape")


def table(src, varname):
    """Pull {"a", "b", "c"} triples out of `var <varname> = []struct{...}{ ... }`."""
    m = re.search(r"var\s+" + varname + r"\s*=\s*\[\]struct\s*\{.*?\}\s*\{(.*?)\n\}", src, re.S)
    assert m, varname
    rows = re.findall(r'\{\s*"([^"]*)"\s*,\s*"([^"]*)"\s*,\s*"([^"]*)"\s*\}', m.group(1))
    return [{"seqOne": a, "seqTwo": b, "aln": go_unquote(c)} for a, b, c in rows]


def local_cases(src, func):
    body = re.search(r"func " + func + r"\(.*?\n\}\n", src, re.S).group(0)
    tg = re.findall(r'(?:tgt|test\.Target)\s*=\s*dna\.StringToBases\("([ACGTN]+)"\)|tgt\s*:=\s*dna\.StringToBases\("([ACGTN]+)"\)', body)
    qy = re.findall(r'(?:qry|test\.Query)\s*=\s*dna\.StringToBases\("([ACGTN]+)"\)|qry\s*:=\s*dna\.StringToBases\("([ACGTN]+)"\)', body)
    pen = re.findall(r"AffineGapLocal\(tgt, qry, DefaultScoreMatrix, (-?\d+), (-?\d+)\)", body)
    eng = re.search(r"GoAffineGapLocalEngine\(DefaultScoreMatrix, (-?\d+), (-?\d+)\)", body)
    exp = re.findall(r'[Ss]core != (-?\d+) \|\| PrintCigar\([a-zA-Z.]+\) != "([0-9MID]+)"', body)
    tg = [a or b for a, b in tg]
    qy = [a or b for a, b in qy]
    assert len(tg) == len(qy) == len(exp), (func, len(tg), len(qy), len(exp))
    out = []
    for idx in range(len(tg)):
        go, ge = pen[idx] if pen else (eng.group(1), eng.group(2))
        out.append({"target": tg[idx], "query": qy[idx], "gapOpen": int(go), "gapExtend": int(ge),
                    "score": int(exp[idx][0]), "cigar": exp[idx][1]})
    return out


def main():
    if not os.path.isdir(REF):
        sys.exit("reference checkout not present; fixtures are already committed")
    aff = open(os.path.join(REF, "align/affineGap_test.go")).read()
    view = open(os.path.join(REF, "align/view_test.go")).read()
    ga = open(os.path.join(REF, "cmd/globalAlignment/globalAlignment_test.go")).read()
    toad = re.search(r'Name: "toad", Seq: dna\.StringToBases\("([ACGT]+)"\)', ga).group(1)
    ahsoka = re.search(r'Name: "ahsoka", Seq: dna\.StringToBases\("([ACGT]+)"\)', ga).group(1)
    tables = {
        "_source": "extracted by tests/golden/make_golden.py from the reference's Go test tables",
        "affineAlignTests": {"matrix": "Default", "gapOpen": -400, "gapExtend": -30,
                             "where": "align/affineGap_test.go:11-25,45-81", "cases": table(aff, "affineAlignTests")},
        "affineAlignChunkTests": {"matrix": "Default", "gapOpen": -400, "gapExtend": -30, "chunkSize": 3,
                                  "where": "align/affineGap_test.go:27-36,83-93", "cases": table(aff, "affineAlignChunkTests")},
        "constAlignTests": {"matrix": "Default", "gapPen": -430,
                            "where": "align/view_test.go:9-38", "cases": table(view, "alignTests")},
        "affineLocalTests": {"matrix": "Default", "where": "align/affineGap_test.go:120-155",
                             "cases": local_cases(aff, "TestAffineGapLocal")},
        "affineLocalEngineTests": {"matrix": "Default", "where": "align/affineGap_test.go:157-192",
                                   "cases": local_cases(aff, "TestGoAffineGapLocalEngine")},
        "globalAlignmentGraph": {"matrix": "HumanChimpTwo", "gapPen": -430,
                                 "where": "cmd/globalAlignment/globalAlignment_test.go:13-40",
                                 "toad": toad, "ahsoka": ahsoka, "expected_nodes": 3},
    }
    assert len(tables["affineAlignTests"]["cases"]) == 9
    assert len(tables["affineAlignChunkTests"]["cases"]) == 4
    assert len(tables["constAlignTests"]["cases"]) == 12
    assert len(tables["affineLocalTests"]["cases"]) == 5
    assert len(tables["affineLocalEngineTests"]["cases"]) == 4
    with open(os.path.join(HERE, "align_tables.json"), "w") as f:
        json.dump(tables, f, indent=1)
        f.write("\n")

    copies = {
        "cmd/globalAlignmentAnchor/testdata": ["hg38.toy.fa", "rheMac10.toy.fa", "out_alignment.1.expected.tsv",
                                               "out_alignment.2.expected.tsv", "out_hg38_gap.1.expected.bed",
                                               "out_hg38_gap.2.expected.bed", "out_rheMac10_gap.1.expected.bed",
                                               "out_rheMac10_gap.2.expected.bed", "out_hg38_alignment.1.expected.bed",
                                               "out_hg38_alignment.2.expected.bed", "out_rheMac10_alignment.1.expected.bed",
                                               "out_rheMac10_alignment.2.expected.bed"],
        "cmd/cigarToBed/testdata/firstTest": ["affineGap_PanTro6vshg38_del.bed", "affineGap_PanTro6vshg38_ins.bed",
                                              "testRegion10kb_PanTro6.fa", "testRegion10kb_hg38.fa"],
        "cmd/cigarToBed/testdata/sethvsraven": ["affineGap_sethvsraven_del.bed", "affineGap_sethvsraven_ins.bed",
                                                "raven.fa", "seth.fa"],
        "cmd/globalAlignment/testdata": ["chelsea.fa", "eric.fa", "faOut_test.fa"],
        "align/testdata": ["multiAlignTest.in.fa", "multiAlignTest.in2.fa", "multiAlignTest.expected.fa", "multiAlignTest.expected2.fa"],
    }
    for d, names in copies.items():
        dst = os.path.join(HERE, "data", d.replace("cmd/", "").replace("/testdata", ""))  # align/testdata -> data/align
        os.makedirs(dst, exist_ok=True)
        for nm in names:
            shutil.copyfile(os.path.join(REF, d, nm), os.path.join(dst, nm))
            os.chmod(os.path.join(dst, nm), 0o644)
    print("golden fixtures written under", HERE)


if __name__ == "__main__":
    main()
