// align_hip.go -- the cgo shim that puts libgonomics_align_hip.so (MI355X) behind gonomics' package align.
//
// Recipe (shim/manifest.json is the machine-readable form, tests/test_shim_static.py checks it against the reference's sources):
// copy shim/align_hip*.go to github.com/vertgenlab/gonomics/align/, give affineGap.go, constGap.go, affineGap_highMem.go and
// constGap_highMem.go the constraint `//go:build !hip`, and build with `go build -tags hip ./...` on a machine with ROCm, the library
// in the linker path and include/gnx_align.h in the include path.  Every package-level name of the four excluded files that a kept
// file (align.go, multiAlign.go, ungapped.go, view.go, draw.go, the *_test.go files) refers to is declared by the shim files.
// No alignment logic here: flatten the score matrix, pass the slices, copy the result out.  NOT COMPILED in the image this
// repository is built in (no Go toolchain); every C entry point it binds is exercised through the same C ABI by tests/ (ctypes)
// and by include/gonomics_align.hpp (C++).
//go:build hip

package align

/*
#cgo LDFLAGS: -lgonomics_align_hip
#include <stdlib.h>
#include "gnx_align.h"
*/
import "C"

import (
	"log"
	"unsafe"

	"github.com/vertgenlab/gonomics/dna"
)

func hipParams(mode C.int32_t, scores [][]int64, gapOpen, gapExtend int64, ci, cj int) C.gnx_params {
	var p C.gnx_params
	p.mode = mode
	for a := 0; a < 5; a++ { // [][]int64 -> row-major int64[25]
		for b := 0; b < 5; b++ {
			p.scores[a*5+b] = C.int64_t(scores[a][b])
		}
	}
	p.gap_open, p.gap_extend = C.int64_t(gapOpen), C.int64_t(gapExtend)
	p.checkersize_i, p.checkersize_j = C.int64_t(ci), C.int64_t(cj)
	return p
}

func basePtr(s []dna.Base) *C.uint8_t { // dna.Base is a byte; no Go pointer is retained by C after return
	if len(s) == 0 {
		return nil
	}
	return (*C.uint8_t)(unsafe.Pointer(&s[0]))
}

// lastError returns the library's error text.  gnx_last_error() falls back to the most recent error of the process when the
// calling OS thread has none, so it is right even if the Go scheduler moved this goroutine between the two cgo calls.
func lastError() string { return C.GoString(C.gnx_last_error()) }

// hipCheck maps a return code onto what the Go code does in the same situation.
func hipCheck(rc C.int) {
	switch rc {
	case C.GNX_OK:
	case C.GNX_EBASE: // a base >= 5 indexes the 5x5 matrix
		panic("runtime error: index out of range (dna.Base used as score-matrix index)")
	case C.GNX_ETRACE: // affineGap.go:335, constGap.go:267,300
		log.Fatalf("Error: unexpected traceback")
	default:
		log.Panicf("align (hip): %s", lastError())
	}
}

// routesFrom copies n CIGARs out of the library's arrays (gnx_cigar has the memory layout of Cigar{RunLength int64; Op ColType})
// and releases them.
func routesFrom(ops *C.gnx_cigar, off *C.int64_t, n int) [][]Cigar {
	defer C.gnx_free(unsafe.Pointer(ops))
	defer C.gnx_free(unsafe.Pointer(off))
	offs := unsafe.Slice(off, n+1)
	all := unsafe.Slice(ops, int(offs[n]))
	routes := make([][]Cigar, n)
	for i := 0; i < n; i++ {
		routes[i] = make([]Cigar, 0, int(offs[i+1]-offs[i]))
		for _, c := range all[offs[i]:offs[i+1]] {
			routes[i] = append(routes[i], Cigar{RunLength: int64(c.run_length), Op: ColType(c.op)})
		}
	}
	return routes
}

func hipPair(p C.gnx_params, alpha, beta []dna.Base) (int64, []Cigar) {
	var score, n C.int64_t
	var ops *C.gnx_cigar
	hipCheck(C.gnx_align_pair(&p, basePtr(alpha), C.int64_t(len(alpha)), basePtr(beta), C.int64_t(len(beta)), &score, &ops, &n))
	defer C.gnx_free(unsafe.Pointer(ops))
	route := make([]Cigar, int(n))
	for i, c := range unsafe.Slice(ops, int(n)) {
		route[i] = Cigar{RunLength: int64(c.run_length), Op: ColType(c.op)}
	}
	return int64(score), route
}

// affineGap.go:59
func AffineGap(alpha []dna.Base, beta []dna.Base, scores [][]int64, gapOpen int64, gapExtend int64) (int64, []Cigar) {
	return AffineGap_customizeCheckersize(alpha, beta, scores, gapOpen, gapExtend, 10000, 10000)
}

// affineGap.go:73
func AffineGap_customizeCheckersize(alpha []dna.Base, beta []dna.Base, scores [][]int64, gapOpen int64, gapExtend int64, checkersize_i int, checkersize_j int) (int64, []Cigar) {
	return hipPair(hipParams(C.GNX_AFFINE_GAP, scores, gapOpen, gapExtend, checkersize_i, checkersize_j), alpha, beta)
}

// constGap.go:13
func ConstGap(alpha []dna.Base, beta []dna.Base, scores [][]int64, gapPen int64) (int64, []Cigar) {
	return ConstGap_customizeCheckersize(alpha, beta, scores, gapPen, 10000, 10000)
}

// constGap.go:73
func ConstGap_customizeCheckersize(alpha []dna.Base, beta []dna.Base, scores [][]int64, gapPen int64, checkersize_i int, checkersize_j int) (int64, []Cigar) {
	return hipPair(hipParams(C.GNX_CONST_GAP, scores, gapPen, 0, checkersize_i, checkersize_j), alpha, beta)
}

// affineGap_highMem.go:99
func AffineGap_highMem(alpha []dna.Base, beta []dna.Base, scores [][]int64, gapOpen int64, gapExtend int64) (int64, []Cigar) {
	return hipPair(hipParams(C.GNX_AFFINE_GAP_HIGHMEM, scores, gapOpen, gapExtend, 10000, 10000), alpha, beta)
}

// affineGap_highMem.go:105
func AffineGapLocal(target []dna.Base, query []dna.Base, scores [][]int64, gapOpen int64, gapExtend int64) (int64, []Cigar) {
	return hipPair(hipParams(C.GNX_AFFINE_GAP_LOCAL, scores, gapOpen, gapExtend, 10000, 10000), target, query)
}

// constGap_highMem.go:11
func ConstGap_highMem(alpha []dna.Base, beta []dna.Base, scores [][]int64, gapPen int64) (int64, []Cigar) {
	return hipPair(hipParams(C.GNX_CONST_GAP_HIGHMEM, scores, gapPen, 0, 10000, 10000), alpha, beta)
}

// TargetQueryPair is the engine's work item (affineGap_highMem.go:110-115): the type leaves with the excluded file, so it lives here.
type TargetQueryPair struct {
	Target []dna.Base
	Query  []dna.Base
	Score  int64
	Cigar  []Cigar
}

// GoAffineGapLocalEngine keeps the channel API (affineGap_highMem.go:120-125): one goroutine drains up to 1000 queued
// pairs, aligns them as one batch and sends the results back in input order (the reference's engine is one worker: FIFO).
func GoAffineGapLocalEngine(scores [][]int64, gapOpen int64, gapExtend int64) (inputs chan<- TargetQueryPair, outputs <-chan TargetQueryPair) {
	in, out := make(chan TargetQueryPair, 1000), make(chan TargetQueryPair, 1000)
	go func() {
		for first := range in {
			batch := []TargetQueryPair{first}
		drain:
			for len(batch) < 1000 {
				select {
				case p, ok := <-in:
					if !ok {
						break drain
					}
					batch = append(batch, p)
				default:
					break drain
				}
			}
			t, q := make([][]dna.Base, len(batch)), make([][]dna.Base, len(batch))
			for i := range batch {
				t[i], q[i] = batch[i].Target, batch[i].Query
			}
			s, r := AlignBatch(int(C.GNX_AFFINE_GAP_LOCAL), t, q, scores, gapOpen, gapExtend, 10000, 10000)
			for i := range batch {
				batch[i].Score, batch[i].Cigar = s[i], r[i]
				out <- batch[i]
			}
		}
		close(out)
	}()
	return in, out
}
