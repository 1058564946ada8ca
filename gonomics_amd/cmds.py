"""Host-side mirrors of the inner functions of the three commands on the path: the callers' I/O either
side of the DP, byte-identical to the reference's golden output files.

  globalAlignment             /root/reference/cmd/globalAlignment/globalAlignment.go:70-108
  GlobalAlignment_CigarToBed  /root/reference/cmd/cigarToBed/cigarToBed.go:65-160
  gapToAlignment              /root/reference/cmd/globalAlignmentAnchor/globalAlignmentAnchor.go:321-436
                              (the serial loop of AffineGap_customizeCheckersize calls becomes ONE batched call)
  faChunkAlign                /root/reference/cmd/faChunkAlign/faChunkAlign.go:18-29

`aligner` parameters default to the GPU path (gonomics_amd.align); tests inject a CPU stand-in to check the I/O logic.
"""
from . import align, dna, fasta


def _read_single(path):
    recs = fasta.Read(path)
    if len(recs) != 1:
        raise SystemExit("multiple sequnces detected in .fa files: this program is designed for .fa files with only 1 sequence in them")
    return recs[0]


def globalAlignment(inputFileOne, inputFileTwo, outFileName="", const_gap=None):
    """Returns the text the Go command prints; writes the two-record MSA fasta when outFileName is given."""
    const_gap = const_gap or align.ConstGap
    faOne, faTwo = _read_single(inputFileOne), _read_single(inputFileTwo)
    bestScore, aln = const_gap(faOne.Seq, faTwo.Seq, align.HumanChimpTwoScoreMatrix, -430)
    visualize = align.View(faOne.Seq, faTwo.Seq, aln)
    out = "Alignment score is %d, cigar is %s \n" % (bestScore, align.FormatCigar(aln)) + visualize + "\n"
    if outFileName:
        v = visualize.split("\n")
        with open(outFileName, "w") as fh:
            fh.write(">" + faOne.Name + "\n" + v[0] + "\n" + ">" + faTwo.Name + "\n" + v[1] + "\n")
    return out


def faChunkAlign(inFile, chunkSize, gapOpen, gapExtend, outFile, all_seq_affine_chunk=None):
    """multi-fasta in -> align.AllSeqAffineChunk(HumanChimpTwoScoreMatrix) -> multi-fasta out (the command negates its -gapOpen /
    -gapExtend flags before this call: pass the negative penalties)"""
    fn = all_seq_affine_chunk or align.AllSeqAffineChunk
    records = fasta.Read(inFile)
    records = fn(records, align.HumanChimpTwoScoreMatrix, gapOpen, gapExtend, chunkSize)
    fasta.Write(outFile, records)
    return records


def GlobalAlignment_CigarToBed(inputFileOne, inputFileTwo, outFa, outIns_bed, outDel_bed, FirstPos_InsBed, FirstPos_DelBed, Chrom, affine_gap=None):
    affine_gap = affine_gap or align.AffineGap
    faOne, faTwo = _read_single(inputFileOne), _read_single(inputFileTwo)
    fasta.ToUpper(faOne)
    fasta.ToUpper(faTwo)
    bestScore, aln = affine_gap(faOne.Seq, faTwo.Seq, align.HumanChimpTwoScoreMatrix, -600, -150)
    out = "Using AffineGap, Alignment score is %d, cigar is %s \n" % (bestScore, align.FormatCigar(aln))
    with open(outIns_bed, "w") as ins:
        cur = FirstPos_InsBed - 1
        for i in range(len(aln) - 1):
            if aln[i].Op == align.ColM and aln[i + 1].Op == align.ColI:
                start = cur + aln[i].RunLength + 1
                ins.write("%s\t%d\t%d\tins\n" % (Chrom, start, start + aln[i + 1].RunLength))
            if aln[i].Op != align.ColD:
                cur += aln[i].RunLength
    with open(outDel_bed, "w") as dele:
        cur = FirstPos_DelBed - 1
        for i in range(len(aln) - 1):
            if aln[i].Op == align.ColM and aln[i + 1].Op == align.ColI:
                start = cur + aln[i].RunLength
                dele.write("%s\t%d\t%d\tdel\n" % (Chrom, start, start + 1))
            if aln[i].Op != align.ColI:
                cur += aln[i].RunLength
    visualize = align.View(faOne.Seq, faTwo.Seq, aln)
    out += visualize + "\n"
    if outFa:
        v = visualize.split("\n")
        with open(outFa, "w") as fh:
            fh.write(">" + faOne.Name + "\n" + v[0] + "\n" + ">" + faTwo.Name + "\n" + v[1] + "\n")
    return out


def read_bed4(path):
    rows = []
    with open(path) as fh:
        for line in fh:
            f = line.rstrip("\n").split("\t")
            if len(f) >= 4:
                rows.append((f[0], int(f[1]), int(f[2]), f[3]))
    return rows


def gapToAlignment(species1_gap_bed, species2_gap_bed, species1_genome, species2_genome, speciesOne, speciesTwo, outFilenamePrefix, align_batch=None):
    """Step 3 of cmd/globalAlignmentAnchor: align every gap pair, write <prefix>.alignment.tsv and the two per-species BEDs.
    The beds are lists of (chrom, start, end, name)."""
    g1 = fasta.ToMap(fasta.Read(species1_genome))
    g2 = fasta.ToMap(fasta.Read(species2_genome))
    todo, alphas, betas = [], [], []
    for i, (b1, b2) in enumerate(zip(species1_gap_bed, species2_gap_bed)):
        if b1[3] == "species1_Insertion" or b2[3] == "species2_Insertion":
            continue
        a = dna.AllToUpper(g1[b1[0]][b1[1] - 1:b1[2] - 1].copy())  # bed [start, end) 1-based -> fasta [start-1, end-1)
        b = dna.AllToUpper(g2[b2[0]][b2[1] - 1:b2[2] - 1].copy())
        todo.append(i); alphas.append(a); betas.append(b)
    if align_batch is None:
        from . import _lib
        params = _lib.make_params(_lib.GNX_AFFINE_GAP, align.HumanChimpTwoScoreMatrix, -600, -150, 10000, 10000)
        results = align.AlignBatch(params, alphas, betas) if todo else []
    else:
        results = align_batch(alphas, betas)
    aligned = dict(zip(todo, results))
    with open(outFilenamePrefix + ".alignment.tsv", "w") as tsv, \
            open(outFilenamePrefix + "_" + speciesOne + "_alignment.bed", "w") as o1, \
            open(outFilenamePrefix + "_" + speciesTwo + "_alignment.bed", "w") as o2:
        bed = lambda r: "%s\t%d\t%d\t%s" % r
        for i, (b1, b2) in enumerate(zip(species1_gap_bed, species2_gap_bed)):
            if b1[3] == "species1_Insertion":
                score = -600 * 1 + (-150) * (b1[2] - b1[1] - 1)
                aln = [align.Cigar(b1[2] - b1[1], align.ColD)]
                tsv.write("%s\t%s\t%d\t%s\n" % (bed(b1), bed(b2), score, align.FormatCigar(aln)))
                o1.write(bed(b1) + "\n")
            elif b2[3] == "species2_Insertion":
                score = -600 * 1 + (-150) * (b2[2] - b2[1] - 1)
                aln = [align.Cigar(b2[2] - b2[1], align.ColI)]
                tsv.write("%s\t%s\t%d\t%s\n" % (bed(b1), bed(b2), score, align.FormatCigar(aln)))
                o2.write(bed(b2) + "\n")
            else:
                score, aln = aligned[i]
                tsv.write("%s\t%s\t%d\t%s\n" % (bed(b1), bed(b2), score, align.FormatCigar(aln)))
                p1, p2 = b1[1], b2[1]
                for c in aln:
                    n = c.RunLength
                    if c.Op == align.ColM:
                        o1.write("%s\t%d\t%d\tspecies1_Match\n" % (b1[0], p1, p1 + n))
                        o2.write("%s\t%d\t%d\tspecies2_Match\n" % (b2[0], p2, p2 + n))
                        p1 += n; p2 += n
                    elif c.Op == align.ColI:
                        o2.write("%s\t%d\t%d\tspecies2_Insertion\n" % (b2[0], p2, p2 + n))
                        p2 += n
                    else:
                        o1.write("%s\t%d\t%d\tspecies1_Insertion\n" % (b1[0], p1, p1 + n))
                        p1 += n
