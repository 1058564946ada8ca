// align_hip_nearest.go -- OPTIONAL second step of the recipe (shim/manifest.json, "step2"): move nearestGroups and
// nearestGroupsChunk (multiAlign.go:27-57) together with multiAlign.go's "math" import -- nothing else in that file uses it -- into a
// file multiAlign_nearest.go tagged `//go:build !hip`, and add this file.  A progressive-alignment round then evaluates all x<y
// group pairs in ONE gnx_multiple_affine_gap_batch call instead of one call per pair; the selection rule is the reference's
// (first strict maximum in x<y order).  Without this step the package builds and runs with the kept multiAlign.go as it is.
//go:build hip

package align

/*
#include "gnx_align.h"
*/
import "C"

import (
	"log"
	"math"

	"github.com/vertgenlab/gonomics/fasta"
)

func nearestGroupsBatched(groups [][]fasta.Fasta, scoreMatrix [][]int64, gapOpen int64, gapExtend int64, chunkSize int64) (bestX int, bestY int, bestScore int64, bestRoute []Cigar) {
	var pairA, pairB []C.int32_t
	for x := 0; x < len(groups)-1; x++ {
		for y := x + 1; y < len(groups); y++ {
			pairA, pairB = append(pairA, C.int32_t(x)), append(pairB, C.int32_t(y))
		}
	}
	scores, routes := multipleGroups(groups, pairA, pairB, scoreMatrix, gapOpen, gapExtend, chunkSize)
	bestScore = math.MinInt64
	for q := range scores {
		if scores[q] > bestScore {
			bestX, bestY, bestScore, bestRoute = int(pairA[q]), int(pairB[q]), scores[q], routes[q]
		}
	}
	return bestX, bestY, bestScore, bestRoute
}

// multiAlign.go:27
func nearestGroups(groups [][]fasta.Fasta, scoreMatrix [][]int64, gapOpen int64, gapExtend int64) (bestX int, bestY int, bestScore int64, bestRoute []Cigar) {
	return nearestGroupsBatched(groups, scoreMatrix, gapOpen, gapExtend, 1)
}

// multiAlign.go:42
func nearestGroupsChunk(groups [][]fasta.Fasta, scoreMatrix [][]int64, gapOpen int64, gapExtend int64, chunkSize int) (bestX int, bestY int, bestScore int64, bestRoute []Cigar) {
	for _, g := range groups { // multipleAffineGapChunk's length checks (affineGap_highMem.go:310-315) on every group a pair would touch
		if len(groups) > 1 && len(g[0].Seq)%chunkSize != 0 {
			log.Fatalf("Error: the first subalignment has a length of %d, when it should be a multiple of %d\n", len(g[0].Seq), chunkSize)
		}
	}
	return nearestGroupsBatched(groups, scoreMatrix, gapOpen, gapExtend, int64(chunkSize))
}
