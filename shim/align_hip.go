// align_hip.go -- the cgo shim that puts libgonomics_align_hip.so (MI355X) behind gonomics' package align.
//
// Copy to github.com/vertgenlab/gonomics/align/align_hip.go and give affineGap.go, constGap.go, affineGap_highMem.go and
// constGap_highMem.go the constraint `//go:build !hip`; build with `go build -tags hip ./...` on a machine with ROCm, the
// library in the linker path and include/gnx_align.h in the include path.  It contains no alignment logic: flatten the score
// matrix, pass the slices, copy the result out.  NOT COMPILED in the image this repository is built in (no Go toolchain there);
// every C entry point it binds is exercised through the same C ABI by tests/ (ctypes) and by include/gonomics_align.hpp (C++).
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

// InitDevices makes every batch entry point below use n GPUs of the node from this one process (0: all visible): the library
// cuts each batch into contiguous blocks of equal DP cells, broadcasts a shared reference over RCCL and gathers in input order.
func InitDevices(n int, workspaceBytesPerDevice int64) {
	if rc := C.gnx_init_devices(C.int(n), nil, C.int64_t(workspaceBytesPerDevice)); rc != C.GNX_OK {
		log.Panicf("align (hip): %s", lastError())
	}
}

// SetReference uploads a genome once; AlignBatchByOffset then aligns reads against windows of it (gsw-style workloads, and
// cmd/globalAlignmentAnchor's loop, which re-slices its two genomes per anchor: globalAlignmentAnchor.go:352-384).
func SetReference(ref []dna.Base) {
	if rc := C.gnx_set_reference(basePtr(ref), C.int64_t(len(ref))); rc != C.GNX_OK {
		log.Panicf("align (hip): %s", lastError())
	}
}

func hipPair(p C.gnx_params, alpha, beta []dna.Base) (int64, []Cigar) {
	var score, n C.int64_t
	var ops *C.gnx_cigar
	rc := C.gnx_align_pair(&p, basePtr(alpha), C.int64_t(len(alpha)), basePtr(beta), C.int64_t(len(beta)), &score, &ops, &n)
	switch rc {
	case C.GNX_OK:
	case C.GNX_EBASE: // what the Go code does on a base >= 5
		panic("runtime error: index out of range (dna.Base used as score-matrix index)")
	case C.GNX_ETRACE:
		log.Fatalf("Error: unexpected traceback")
	default:
		log.Panicf("align (hip): %s", lastError())
	}
	defer C.gnx_free(unsafe.Pointer(ops))
	route := make([]Cigar, int(n)) // gnx_cigar has the memory layout of Cigar{RunLength int64; Op ColType}
	src := unsafe.Slice((*C.gnx_cigar)(unsafe.Pointer(ops)), int(n))
	for i := range route {
		route[i] = Cigar{RunLength: int64(src[i].run_length), Op: ColType(src[i].op)}
	}
	return int64(score), route
}

func AffineGap(alpha, beta []dna.Base, scores [][]int64, gapOpen, gapExtend int64) (int64, []Cigar) {
	return AffineGap_customizeCheckersize(alpha, beta, scores, gapOpen, gapExtend, 10000, 10000)
}
func AffineGap_customizeCheckersize(alpha, beta []dna.Base, scores [][]int64, gapOpen, gapExtend int64, ci, cj int) (int64, []Cigar) {
	return hipPair(hipParams(C.GNX_AFFINE_GAP, scores, gapOpen, gapExtend, ci, cj), alpha, beta)
}
func ConstGap(alpha, beta []dna.Base, scores [][]int64, gapPen int64) (int64, []Cigar) {
	return ConstGap_customizeCheckersize(alpha, beta, scores, gapPen, 10000, 10000)
}
func ConstGap_customizeCheckersize(alpha, beta []dna.Base, scores [][]int64, gapPen int64, ci, cj int) (int64, []Cigar) {
	return hipPair(hipParams(C.GNX_CONST_GAP, scores, gapPen, 0, ci, cj), alpha, beta)
}
func AffineGap_highMem(alpha, beta []dna.Base, scores [][]int64, gapOpen, gapExtend int64) (int64, []Cigar) {
	return hipPair(hipParams(C.GNX_AFFINE_GAP_HIGHMEM, scores, gapOpen, gapExtend, 10000, 10000), alpha, beta)
}
func AffineGapLocal(target, query []dna.Base, scores [][]int64, gapOpen, gapExtend int64) (int64, []Cigar) {
	return hipPair(hipParams(C.GNX_AFFINE_GAP_LOCAL, scores, gapOpen, gapExtend, 10000, 10000), target, query)
}
func ConstGap_highMem(alpha, beta []dna.Base, scores [][]int64, gapPen int64) (int64, []Cigar) {
	return hipPair(hipParams(C.GNX_CONST_GAP_HIGHMEM, scores, gapPen, 0, 10000, 10000), alpha, beta)
}

// AlignBatch is the batched form for loops over independent pairs (cmd/globalAlignmentAnchor.go:352-384).
func AlignBatch(mode int, alphas, betas [][]dna.Base, scores [][]int64, gapOpen, gapExtend int64, ci, cj int) ([]int64, [][]Cigar) {
	n := len(alphas)
	aOff, bOff := make([]C.int64_t, n+1), make([]C.int64_t, n+1)
	var aCat, bCat []dna.Base
	for i := 0; i < n; i++ {
		aCat, bCat = append(aCat, alphas[i]...), append(bCat, betas[i]...)
		aOff[i+1], bOff[i+1] = C.int64_t(len(aCat)), C.int64_t(len(bCat))
	}
	p := hipParams(C.int32_t(mode), scores, gapOpen, gapExtend, ci, cj)
	out := make([]int64, n)
	var ops *C.gnx_cigar
	var off *C.int64_t
	if rc := C.gnx_align_batch(&p, C.int64_t(n), basePtr(aCat), &aOff[0], basePtr(bCat), &bOff[0],
		(*C.int64_t)(unsafe.Pointer(&out[0])), &ops, &off); rc != C.GNX_OK {
		log.Panicf("align (hip): %s", lastError())
	}
	defer C.gnx_free(unsafe.Pointer(ops))
	defer C.gnx_free(unsafe.Pointer(off))
	offs := unsafe.Slice(off, n+1)
	all := unsafe.Slice(ops, int(offs[n]))
	routes := make([][]Cigar, n)
	for i := 0; i < n; i++ {
		for _, c := range all[offs[i]:offs[i+1]] {
			routes[i] = append(routes[i], Cigar{RunLength: int64(c.run_length), Op: ColType(c.op)})
		}
	}
	return out, routes
}

// AlignBatchByOffset: reads against windows (start, length) of the resident reference.
func AlignBatchByOffset(mode int, reads [][]dna.Base, refStart, refLen []int64, scores [][]int64, gapOpen, gapExtend int64) ([]int64, [][]Cigar) {
	n := len(reads)
	aOff := make([]C.int64_t, n+1)
	var aCat []dna.Base
	for i := 0; i < n; i++ {
		aCat = append(aCat, reads[i]...)
		aOff[i+1] = C.int64_t(len(aCat))
	}
	p := hipParams(C.int32_t(mode), scores, gapOpen, gapExtend, 10000, 10000)
	out := make([]int64, n)
	var ops *C.gnx_cigar
	var off *C.int64_t
	if rc := C.gnx_align_batch_by_offset(&p, C.int64_t(n), basePtr(aCat), &aOff[0], (*C.int64_t)(unsafe.Pointer(&refStart[0])),
		(*C.int64_t)(unsafe.Pointer(&refLen[0])), (*C.int64_t)(unsafe.Pointer(&out[0])), &ops, &off); rc != C.GNX_OK {
		log.Panicf("align (hip): %s", lastError())
	}
	defer C.gnx_free(unsafe.Pointer(ops))
	defer C.gnx_free(unsafe.Pointer(off))
	offs := unsafe.Slice(off, n+1)
	all := unsafe.Slice(ops, int(offs[n]))
	routes := make([][]Cigar, n)
	for i := 0; i < n; i++ {
		for _, c := range all[offs[i]:offs[i+1]] {
			routes[i] = append(routes[i], Cigar{RunLength: int64(c.run_length), Op: ColType(c.op)})
		}
	}
	return out, routes
}

// GoAffineGapLocalEngine keeps the channel API (affineGap_highMem.go:120-125): one goroutine drains up to 1000 queued
// pairs, aligns them as one batch and sends the results back in input order.
func GoAffineGapLocalEngine(scores [][]int64, gapOpen, gapExtend int64) (chan<- TargetQueryPair, <-chan TargetQueryPair) {
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
