// dynamicAln_hip.go -- the seed-extension DPs of the graph aligner (cmd/gsw) behind libgonomics_align_hip.so.
//
// Recipe (shim/manifest.json, package genomeGraph): move LeftDynamicAln and RightDynamicAln (genomeGraph/search.go:234-321) --
// the two functions only, text unchanged -- into a new file genomeGraph/dynamicAln.go tagged `//go:build !hip`, and add this file.
// Both keep their signatures, so LeftAlignTraversal / RightAlignTraversal (search.go:166-232) and everything above them compile
// unchanged.  *MatrixAln is accepted and not used (the DP matrix lives on the device).  What stays in Go is the route-building
// loop of search.go:252-262 / 298-308: resetDynamicScore (search.go:104-107) receives its argument BY VALUE and resets nothing,
// so a route handed in by a sibling branch is kept and merged with a routeIdx that restarts at 0 -- mergeRoute replays exactly
// that over the traced runs.  dynamicScore.currMax is 0 in every caller (routines.go:18-56 create the keeper empty and nothing
// can write to it through the by-value parameters); a non-zero value is refused rather than silently ignored.
// One call = one extension; GswExtendBatch is the batched form a batched GraphSmithWatermanToGiraf uses (INTEGRATION.md).
// NOT COMPILED in the image this repository is built in (no Go toolchain).
//go:build hip

package genomeGraph

/*
#cgo LDFLAGS: -lgonomics_align_hip
#include <stdlib.h>
#include "gnx_align.h"
*/
import "C"

import (
	"log"
	"unsafe"

	"github.com/vertgenlab/gonomics/cigar"
	"github.com/vertgenlab/gonomics/dna"
)

var colToOp = [3]byte{cigar.Match, cigar.Insertion, cigar.Deletion} // GNX_COL_M / I / D

func gswBasePtr(s []dna.Base) *C.uint8_t {
	if len(s) == 0 {
		return nil
	}
	return (*C.uint8_t)(unsafe.Pointer(&s[0]))
}

// mergeRoute is the loop of search.go:252-262 (== 298-308) over runs in traceback order.
func mergeRoute(route []cigar.Cigar, runs []C.gnx_cigar) []cigar.Cigar {
	routeIdx := 0
	for _, r := range runs {
		op := colToOp[r.op]
		for k := int64(0); k < int64(r.run_length); k++ {
			if len(route) == 0 {
				route = append(route, cigar.Cigar{RunLength: 1, Op: op})
			} else if route[routeIdx].Op == op {
				route[routeIdx].RunLength += 1
			} else {
				route = append(route, cigar.Cigar{RunLength: 1, Op: op})
				routeIdx++
			}
		}
	}
	return route
}

// GswExtendBatch runs one side (C.GNX_GSW_LEFT / C.GNX_GSW_RIGHT) of a batch of extensions: alphas[i] against betas[i].
// routesIn[i] (may be nil) is the route the reference would have found in dynamicScore.route.
func GswExtendBatch(side int, alphas, betas [][]dna.Base, scores [][]int64, gapPen int64, routesIn [][]cigar.Cigar) ([]int64, [][]cigar.Cigar, []int, []int) {
	n := len(alphas)
	if n == 0 {
		return nil, nil, nil, nil
	}
	var flat [25]C.int64_t
	for a := 0; a < 5; a++ {
		for b := 0; b < 5; b++ {
			flat[a*5+b] = C.int64_t(scores[a][b])
		}
	}
	aOff, bOff := make([]C.int64_t, n+1), make([]C.int64_t, n+1)
	var aCat, bCat []dna.Base
	for i := 0; i < n; i++ {
		aCat, bCat = append(aCat, alphas[i]...), append(bCat, betas[i]...)
		aOff[i+1], bOff[i+1] = C.int64_t(len(aCat)), C.int64_t(len(bCat))
	}
	score, endI, endJ := make([]C.int64_t, n), make([]C.int64_t, n), make([]C.int64_t, n)
	var ops *C.gnx_cigar
	var off *C.int64_t
	rc := C.gnx_gsw_extend_batch(C.int(side), &flat[0], C.int64_t(gapPen), C.int64_t(n), gswBasePtr(aCat), &aOff[0], gswBasePtr(bCat), &bOff[0],
		&score[0], &endI[0], &endJ[0], &ops, &off)
	switch rc {
	case C.GNX_OK:
	case C.GNX_EBASE:
		panic("runtime error: index out of range (dna.Base used as score-matrix index)")
	default:
		log.Panicf("genomeGraph (hip): %s", C.GoString(C.gnx_last_error()))
	}
	defer C.gnx_free(unsafe.Pointer(ops))
	defer C.gnx_free(unsafe.Pointer(off))
	offs := unsafe.Slice(off, n+1)
	all := unsafe.Slice(ops, int(offs[n]))
	outScore, outRoute, outI, outJ := make([]int64, n), make([][]cigar.Cigar, n), make([]int, n), make([]int, n)
	for i := 0; i < n; i++ {
		var in []cigar.Cigar
		if routesIn != nil {
			in = routesIn[i]
		}
		outScore[i], outRoute[i], outI[i], outJ[i] = int64(score[i]), mergeRoute(in, all[offs[i]:offs[i+1]]), int(endI[i]), int(endJ[i])
	}
	return outScore, outRoute, outI, outJ
}

func dynamicAlnOne(side int, alpha []dna.Base, beta []dna.Base, scores [][]int64, gapPen int64, dynamicScore dynamicScoreKeeper) (int64, []cigar.Cigar, int, int) {
	if dynamicScore.currMax != 0 {
		log.Panicf("genomeGraph (hip): dynamicScore.currMax = %d: no caller of the reference passes a non-zero value", dynamicScore.currMax)
	}
	s, r, i, j := GswExtendBatch(side, [][]dna.Base{alpha}, [][]dna.Base{beta}, scores, gapPen, [][]cigar.Cigar{dynamicScore.route})
	return s[0], r[0], i[0], j[0]
}

// search.go:234
func LeftDynamicAln(alpha []dna.Base, beta []dna.Base, scores [][]int64, matrix *MatrixAln, gapPen int64, dynamicScore dynamicScoreKeeper) (int64, []cigar.Cigar, int, int) {
	return dynamicAlnOne(C.GNX_GSW_LEFT, alpha, beta, scores, gapPen, dynamicScore)
}

// search.go:278
func RightDynamicAln(alpha []dna.Base, beta []dna.Base, scores [][]int64, matrix *MatrixAln, gapPen int64, dynamicScore dynamicScoreKeeper) (int64, []cigar.Cigar, int, int) {
	return dynamicAlnOne(C.GNX_GSW_RIGHT, alpha, beta, scores, gapPen, dynamicScore)
}
