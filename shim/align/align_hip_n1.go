// align_hip_n1.go -- the chunk / multiple-alignment functions that leave with affineGap_highMem.go under the `hip` tag
// (what cmd/faChunkAlign and popgen/dunn.go run through AllSeqAffine / AllSeqAffineChunk, multiAlign.go:59-78).  The kept
// multiAlign.go:27-57 (nearestGroups, nearestGroupsChunk) calls multipleAffineGap / multipleAffineGapChunk and compiles against
// these as they are; align_hip_nearest.go is the optional second step that evaluates a whole round in one call.
//go:build hip

package align

/*
#include "gnx_align.h"
*/
import "C"

import (
	"fmt"
	"log"
	"unsafe"

	"github.com/vertgenlab/gonomics/dna"
	"github.com/vertgenlab/gonomics/fasta"
)

// affineGap_highMem.go:227 -- the affine DP over chunks of chunkSize bases; run lengths come back in bases.
func AffineGapChunk(alpha []dna.Base, beta []dna.Base, scores [][]int64, gapOpen int64, gapExtend int64, chunkSize int64) (int64, []Cigar) {
	if int64(len(alpha))%chunkSize != 0 { // affineGap_highMem.go:229-234
		log.Fatalf("Error: the first sequence, %s, has a length of %d, when it should be a multiple of %d\n", dna.BasesToString(alpha), len(alpha), chunkSize)
	}
	if int64(len(beta))%chunkSize != 0 {
		log.Fatalf("Error: the second sequence, %s, has a length of %d, when it should be a multiple of %d\n", dna.BasesToString(beta), len(beta), chunkSize)
	}
	p := hipParams(C.GNX_AFFINE_GAP_HIGHMEM, scores, gapOpen, gapExtend, 10000, 10000)
	aOff := [2]C.int64_t{0, C.int64_t(len(alpha))}
	bOff := [2]C.int64_t{0, C.int64_t(len(beta))}
	var score C.int64_t
	var ops *C.gnx_cigar
	var off *C.int64_t
	hipCheck(C.gnx_affine_gap_chunk_batch(&p, C.int64_t(chunkSize), 1, basePtr(alpha), &aOff[0], basePtr(beta), &bOff[0], &score, &ops, &off))
	return int64(score), routesFrom(ops, off, 1)[0]
}

// flattenGroups lays alignment blocks out as gnx_multiple_affine_gap_batch wants them: sequence-major bases, n+1 offsets,
// sequences per group, columns per group.  Every sequence of a group must have the group's length (the Go code indexes
// alpha[k].Seq[col] for col < len(alpha[0].Seq) and panics otherwise).
func flattenGroups(groups [][]fasta.Fasta) ([]dna.Base, []C.int64_t, []C.int32_t, []C.int64_t) {
	off, nseq, glen := make([]C.int64_t, len(groups)+1), make([]C.int32_t, len(groups)), make([]C.int64_t, len(groups))
	var cat []dna.Base
	for g, grp := range groups {
		nseq[g], glen[g] = C.int32_t(len(grp)), C.int64_t(len(grp[0].Seq))
		for _, rec := range grp {
			if len(rec.Seq) != len(grp[0].Seq) {
				panic("runtime error: index out of range (alignment block with rows of unequal length)")
			}
			cat = append(cat, rec.Seq...)
		}
		off[g+1] = C.int64_t(len(cat))
	}
	return cat, off, nseq, glen
}

// multipleGroups runs the given (a, b) group pairs in one call.
func multipleGroups(groups [][]fasta.Fasta, pairA, pairB []C.int32_t, scores [][]int64, gapOpen, gapExtend, chunkSize int64) ([]int64, [][]Cigar) {
	n := len(pairA)
	if n == 0 {
		return nil, nil
	}
	cat, off, nseq, glen := flattenGroups(groups)
	p := hipParams(C.GNX_AFFINE_GAP_HIGHMEM, scores, gapOpen, gapExtend, 10000, 10000)
	out := make([]int64, n)
	var ops *C.gnx_cigar
	var ooff *C.int64_t
	rc := C.gnx_multiple_affine_gap_batch(&p, C.int64_t(chunkSize), C.int64_t(len(groups)), basePtr(cat), &off[0], &nseq[0], &glen[0],
		C.int64_t(n), &pairA[0], &pairB[0], (*C.int64_t)(unsafe.Pointer(&out[0])), &ops, &ooff)
	if rc == C.GNX_EDIVZERO { // a column pair of gaps only: sum / count with count == 0 (multiAlign.go:101)
		panic("runtime error: integer divide by zero")
	}
	hipCheck(rc)
	return out, routesFrom(ops, ooff, n)
}

// affineGap_highMem.go:274
func multipleAffineGap(alpha []fasta.Fasta, beta []fasta.Fasta, scores [][]int64, gapOpen int64, gapExtend int64) (int64, []Cigar) {
	s, r := multipleGroups([][]fasta.Fasta{alpha, beta}, []C.int32_t{0}, []C.int32_t{1}, scores, gapOpen, gapExtend, 1)
	return s[0], r[0]
}

// affineGap_highMem.go:308
func multipleAffineGapChunk(alpha []fasta.Fasta, beta []fasta.Fasta, scores [][]int64, gapOpen int64, gapExtend int64, chunkSize int64) (int64, []Cigar) {
	if int64(len(alpha[0].Seq))%chunkSize != 0 { // affineGap_highMem.go:310-315
		log.Fatalf("Error: the first subalignment has a length of %d, when it should be a multiple of %d\n", len(alpha[0].Seq), chunkSize)
	}
	if int64(len(beta[0].Seq))%chunkSize != 0 {
		log.Fatalf("Error: the second subalignment has a length of %d, when it should be a multiple of %d\n", len(beta[0].Seq), chunkSize)
	}
	s, r := multipleGroups([][]fasta.Fasta{alpha, beta}, []C.int32_t{0}, []C.int32_t{1}, scores, gapOpen, gapExtend, chunkSize)
	return s[0], r[0]
}

// scoreAffineAln (affineGap_highMem.go:355, used by TestAffineScore only) scores a finished two-row alignment; host arithmetic
// over the alignment columns, no DP: every column of two bases adds its matrix entry, every gap column adds gapExtend and the
// first column of a gap run gapOpen as well, separately for the two rows.
func scoreAffineAln(alpha fasta.Fasta, beta fasta.Fasta, scores [][]int64, gapOpen int64, gapExtend int64) (int64, error) {
	if len(alpha.Seq) != len(beta.Seq) {
		return 0, fmt.Errorf("Error: alignment being scored has sequences of unequal length: %d, %d\n", len(alpha.Seq), len(beta.Seq))
	}
	var total int64
	prevGap := [2]bool{false, false}
	for col := range alpha.Seq {
		rows := [2]dna.Base{alpha.Seq[col], beta.Seq[col]}
		if rows[0] != dna.Gap && rows[1] != dna.Gap {
			total += scores[rows[0]][rows[1]]
		}
		for r := 0; r < 2; r++ {
			isGap := rows[r] == dna.Gap
			if isGap {
				total += gapExtend
				if !prevGap[r] {
					total += gapOpen
				}
			}
			prevGap[r] = isGap
		}
	}
	return total, nil
}
