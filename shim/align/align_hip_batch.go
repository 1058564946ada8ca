// align_hip_batch.go -- NEW exported functions of package align under the `hip` tag: the batched forms a caller moves its loop
// over independent pairs to (cmd/globalAlignmentAnchor/globalAlignmentAnchor.go:352-384), the resident reference for gsw-style
// workloads, and the multi-GPU switch.  Nothing in the reference corresponds to them; the Go-signature functions of align_hip.go
// do not need them (GoAffineGapLocalEngine uses AlignBatch).
//go:build hip

package align

/*
#include "gnx_align.h"
*/
import "C"

import (
	"log"
	"unsafe"

	"github.com/vertgenlab/gonomics/dna"
)

// Transport names what carried the last multi-GPU broadcast / gather (gnx_timing.transport): 0 one device, 1 RCCL over xGMI,
// 2 peer copies (GNX_RCCL=0 or the same device listed twice), 3 peer copies after a RCCL call failed.
func Transport() int {
	var t C.gnx_timing
	if C.gnx_get_timing(&t) != C.GNX_OK {
		return -1
	}
	return int(t.transport)
}

// InitDevices makes every batch entry point below use n GPUs of the node from this one process (0: all visible): the library
// cuts each batch into contiguous blocks of equal DP cells, broadcasts a shared reference over RCCL and gathers in input order.
// Call order relative to SetReference does not matter: a resident reference is re-broadcast to contexts that lack it.
func InitDevices(n int, workspaceBytesPerDevice int64) {
	hipCheck(C.gnx_init_devices(C.int(n), nil, C.int64_t(workspaceBytesPerDevice)))
}

// SetReference uploads a genome once (kept on the device 2 bits per base + an N mask); AlignBatchByOffset then aligns reads
// against windows of it.
func SetReference(ref []dna.Base) {
	hipCheck(C.gnx_set_reference(basePtr(ref), C.int64_t(len(ref))))
}

// catBases concatenates a batch for the C ABI: bases and n+1 offsets.
func catBases(seqs [][]dna.Base) ([]dna.Base, []C.int64_t) {
	off := make([]C.int64_t, len(seqs)+1)
	total := 0
	for _, s := range seqs {
		total += len(s)
	}
	cat := make([]dna.Base, 0, total)
	for i, s := range seqs {
		cat = append(cat, s...)
		off[i+1] = C.int64_t(len(cat))
	}
	return cat, off
}

// AlignBatch aligns alphas[i] against betas[i] for every i in one call; mode is one of the gnx_mode values (0 AffineGap,
// 1 ConstGap, 2 AffineGap_highMem, 3 AffineGapLocal, 4 ConstGap_highMem).  Results are in input order.
func AlignBatch(mode int, alphas, betas [][]dna.Base, scores [][]int64, gapOpen, gapExtend int64, ci, cj int) ([]int64, [][]Cigar) {
	n := len(alphas)
	if n != len(betas) {
		log.Panicf("align (hip): AlignBatch: %d alphas, %d betas", n, len(betas))
	}
	if n == 0 {
		return nil, nil
	}
	aCat, aOff := catBases(alphas)
	bCat, bOff := catBases(betas)
	p := hipParams(C.int32_t(mode), scores, gapOpen, gapExtend, ci, cj)
	out := make([]int64, n)
	var ops *C.gnx_cigar
	var off *C.int64_t
	hipCheck(C.gnx_align_batch(&p, C.int64_t(n), basePtr(aCat), &aOff[0], basePtr(bCat), &bOff[0],
		(*C.int64_t)(unsafe.Pointer(&out[0])), &ops, &off))
	return out, routesFrom(ops, off, n)
}

// AlignBatchByOffset: reads[i] against reference[refStart[i] : refStart[i]+refLen[i]] of the resident reference.
func AlignBatchByOffset(mode int, reads [][]dna.Base, refStart, refLen []int64, scores [][]int64, gapOpen, gapExtend int64) ([]int64, [][]Cigar) {
	n := len(reads)
	if n != len(refStart) || n != len(refLen) {
		log.Panicf("align (hip): AlignBatchByOffset: %d reads, %d starts, %d lengths", n, len(refStart), len(refLen))
	}
	if n == 0 {
		return nil, nil
	}
	aCat, aOff := catBases(reads)
	p := hipParams(C.int32_t(mode), scores, gapOpen, gapExtend, 10000, 10000)
	out := make([]int64, n)
	var ops *C.gnx_cigar
	var off *C.int64_t
	hipCheck(C.gnx_align_batch_by_offset(&p, C.int64_t(n), basePtr(aCat), &aOff[0], (*C.int64_t)(unsafe.Pointer(&refStart[0])),
		(*C.int64_t)(unsafe.Pointer(&refLen[0])), (*C.int64_t)(unsafe.Pointer(&out[0])), &ops, &off))
	return out, routesFrom(ops, off, n)
}
