// routines_hip.go -- the graph aligner's worker routine (cmd/gsw) as batches behind libgonomics_align_hip.so.
//
// Recipe (shim/manifest.json, package genomeGraph, step2 -- optional, on top of step1): add this file.  Nothing of the package is
// replaced: RoutineFqToGirafHip is a worker with the contract of RoutineFqToGiraf (genomeGraph/routines.go:12-25: reads off a
// channel, one giraf.Giraf per read onto a channel, wg.Done() at the end) that a cmd/gsw built with `-tags hip` starts ONCE instead
// of `-t` copies of RoutineFqToGiraf: it cuts the stream into batches and hands each to gnx_gsw_map_reads, where the seed search and
// the extension DPs of the whole batch run on the device and the per-read bookkeeping of GraphSmithWatermanToGiraf
// (toGiraf.go:17-72) on a pool of host threads inside the library.  Girafs leave in input order.
// A read on which GraphSmithWatermanToGiraf panics (getLeftTargetBases with a short Prev node, search.go:139) panics here too.
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
	"sync"
	"unsafe"

	"github.com/vertgenlab/gonomics/cigar"
	"github.com/vertgenlab/gonomics/dna"
	"github.com/vertgenlab/gonomics/fastq"
	"github.com/vertgenlab/gonomics/giraf"
)

// HipGraph is a genome graph and its seed index (IndexGenomeIntoMap, index.go:21-59) kept by the library, the index resident on the device.
type HipGraph struct {
	h *C.gnx_gsw_graph
}

// hipEdge is one edge of the graph as AddEdge(u, v) would add it; occ counts parallel edges u -> v (the k-th such entry of u.Next is the k-th of v.Prev).
type hipEdge struct {
	u, v uint32
	occ  int
}

// hipEdgeOrder returns the edges of gg in AN order of AddEdge calls (genomeGraph.go:118-121) that rebuilds every node's Next AND Prev list
// exactly as they are.  The traversals try a node's branches in list order (RightAlignTraversal: Next, LeftAlignTraversal: Prev, search.go:166-232),
// so both orders are part of the graph -- and the reference's own constructors do not add edges node by node: VariantGraph calls
// AddEdge(altAllele, currMatch) before AddEdge(refAllele, currMatch) (graphTools.go:100-108), so the match node after a SNP has Prev = [k+2, k+1].
// The order is a topological order of the edges under "before its successor in u's Next list" and "before its successor in v's Prev list"
// (the smallest ready edge first: graphs built node by node come back in their original order).  Lists that no sequence of AddEdge calls
// produces (a Prev entry without its Next entry, or a cycle between the two orders) panic: the library could not rebuild them.
func hipEdgeOrder(gg *GenomeGraph) []hipEdge {
	var edges []hipEdge
	slot := make(map[hipEdge]int)
	for i := range gg.Nodes {
		seen := make(map[uint32]int)
		for _, e := range gg.Nodes[i].Next {
			seen[e.Dest.Id]++
			k := hipEdge{gg.Nodes[i].Id, e.Dest.Id, seen[e.Dest.Id]}
			slot[k] = len(edges)
			edges = append(edges, k)
		}
	}
	succ := make([][]int, len(edges))
	indeg := make([]int, len(edges))
	for x := 1; x < len(edges); x++ { // the order inside one node's Next list
		if edges[x].u == edges[x-1].u {
			succ[x-1] = append(succ[x-1], x)
			indeg[x]++
		}
	}
	nPrev := 0
	for i := range gg.Nodes {
		seen := make(map[uint32]int)
		last := -1
		for _, e := range gg.Nodes[i].Prev {
			seen[e.Dest.Id]++
			x, ok := slot[hipEdge{e.Dest.Id, gg.Nodes[i].Id, seen[e.Dest.Id]}]
			if !ok {
				log.Panicf("genomeGraph (hip): the Prev edge %d -> %d has no Next edge", e.Dest.Id, gg.Nodes[i].Id)
			}
			if last >= 0 {
				succ[last] = append(succ[last], x)
				indeg[x]++
			}
			last = x
			nPrev++
		}
	}
	if nPrev != len(edges) {
		log.Panicf("genomeGraph (hip): the Next lists hold %d edges, the Prev lists %d", len(edges), nPrev)
	}
	// Kahn's algorithm with the smallest ready edge first (a binary min-heap of edge indices)
	var heap []int
	push := func(x int) {
		heap = append(heap, x)
		for c := len(heap) - 1; c > 0; {
			p := (c - 1) / 2
			if heap[p] <= heap[c] {
				break
			}
			heap[p], heap[c] = heap[c], heap[p]
			c = p
		}
	}
	pop := func() int {
		top := heap[0]
		n := len(heap) - 1
		heap[0] = heap[n]
		heap = heap[:n]
		for p := 0; ; {
			c := 2*p + 1
			if c >= n {
				break
			}
			if c+1 < n && heap[c+1] < heap[c] {
				c++
			}
			if heap[p] <= heap[c] {
				break
			}
			heap[p], heap[c] = heap[c], heap[p]
			p = c
		}
		return top
	}
	for x := range edges {
		if indeg[x] == 0 {
			push(x)
		}
	}
	out := make([]hipEdge, 0, len(edges))
	for len(heap) > 0 {
		x := pop()
		out = append(out, edges[x])
		for _, y := range succ[x] {
			indeg[y]--
			if indeg[y] == 0 {
				push(y)
			}
		}
	}
	if len(out) != len(edges) {
		log.Panicf("genomeGraph (hip): no order of AddEdge calls builds these Next and Prev lists (%d of %d edges ordered)", len(out), len(edges))
	}
	return out
}

// NewHipGraph hands the nodes and the edges of gg to the library, the edges in an order of AddEdge calls that rebuilds the Next AND the
// Prev lists as they are (hipEdgeOrder): in-memory graphs of VariantGraph (a SNP: Prev = [alt, ref]) or of cmd/cigarToBed-style builders
// that add (i-1 -> i) before (i-2 -> i) align with the reference's branch order (ADVICE r5; before: only graphs whose Prev lists were in
// the order of the Next lists -- what Read builds -- were accepted).
func NewHipGraph(gg *GenomeGraph, seedLen int, stepSize int) *HipGraph {
	off := make([]C.int64_t, len(gg.Nodes)+1)
	var cat []dna.Base
	for i := range gg.Nodes {
		if gg.Nodes[i].Id != uint32(i) {
			log.Panicf("genomeGraph (hip): node %d has Id %d (the library addresses nodes by their index)", i, gg.Nodes[i].Id)
		}
		cat = append(cat, gg.Nodes[i].Seq...)
		off[i+1] = C.int64_t(len(cat))
	}
	order := hipEdgeOrder(gg)
	from, to := make([]C.int32_t, len(order)), make([]C.int32_t, len(order))
	for k, e := range order {
		from[k], to[k] = C.int32_t(e.u), C.int32_t(e.v)
	}
	var fp, tp *C.int32_t
	if len(from) > 0 {
		fp, tp = &from[0], &to[0]
	}
	var h *C.gnx_gsw_graph
	if rc := C.gnx_gsw_graph_create(gswBasePtr(cat), &off[0], C.int64_t(len(gg.Nodes)), fp, tp, C.int64_t(len(from)), C.int(seedLen), C.int(stepSize), &h); rc != C.GNX_OK {
		log.Panicf("genomeGraph (hip): %s", C.GoString(C.gnx_last_error()))
	}
	return &HipGraph{h: h}
}

// Close releases the graph and its index.
func (g *HipGraph) Close() {
	C.gnx_gsw_graph_free(g.h)
	g.h = nil
}

const gswPanicText = "runtime error: slice bounds out of range (getLeftTargetBases, search.go:139)"

// GswBatchToGiraf is GraphSmithWatermanToGiraf (toGiraf.go:17-72) for every read of a batch.  threads = 0: the library's default.
// Panics like the reference on the first read it would have panicked on.
func (g *HipGraph) GswBatchToGiraf(reads []fastq.FastqBig, scoreMatrix [][]int64, threads int) []giraf.Giraf {
	out, bad := g.gswBatch(reads, scoreMatrix, threads)
	if bad >= 0 {
		panic(gswPanicText)
	}
	return out
}

// gswBatch: the girafs of the reads before the first one GraphSmithWatermanToGiraf panics on, and that read's index (-1: none)
func (g *HipGraph) gswBatch(reads []fastq.FastqBig, scoreMatrix [][]int64, threads int) ([]giraf.Giraf, int) {
	n := len(reads)
	if n == 0 {
		return nil, -1
	}
	var flat [25]C.int64_t
	for a := 0; a < 5; a++ {
		for b := 0; b < 5; b++ {
			flat[a*5+b] = C.int64_t(scoreMatrix[a][b])
		}
	}
	off := make([]C.int64_t, n+1)
	var cat []dna.Base
	for i := 0; i < n; i++ {
		cat = append(cat, reads[i].Seq...)
		off[i+1] = C.int64_t(len(cat))
	}
	var gir *C.gnx_giraf
	var nodes *C.uint32_t
	var cig *C.gnx_cigar
	rc := C.gnx_gsw_map_reads(g.h, gswBasePtr(cat), &off[0], C.int64_t(n), 0, &flat[0], -600, C.int(threads), &gir, &nodes, &cig)
	switch rc {
	case C.GNX_OK:
	case C.GNX_EBASE:
		panic("runtime error: index out of range (dna.Base used as score-matrix index)")
	default:
		log.Panicf("genomeGraph (hip): %s", C.GoString(C.gnx_last_error()))
	}
	defer C.gnx_free(unsafe.Pointer(gir))
	defer C.gnx_free(unsafe.Pointer(nodes))
	defer C.gnx_free(unsafe.Pointer(cig))
	recs := unsafe.Slice(gir, n)
	last := recs[n-1]
	atLeastOne := func(k int) int { // (the library allocates max(count, 1) elements)
		if k < 1 {
			return 1
		}
		return k
	}
	allNodes := unsafe.Slice(nodes, atLeastOne(int(last.node_off+last.n_nodes)))
	allCig := unsafe.Slice(cig, atLeastOne(int(last.cigar_off+last.n_cigar)))
	out := make([]giraf.Giraf, 0, n)
	for i := 0; i < n; i++ {
		r := recs[i]
		if r.panicked != 0 { // RoutineFqToGiraf would already have sent the girafs of the reads before this one (routines.go:17-21)
			return out, i
		}
		path := giraf.Path{TStart: int(r.t_start), TEnd: int(r.t_end)} // Nodes stays nil for a read without a hit (toGiraf.go:22)
		if r.n_nodes > 0 {
			path.Nodes = make([]uint32, int(r.n_nodes))
		}
		for k := range path.Nodes {
			path.Nodes[k] = uint32(allNodes[int(r.node_off)+k])
		}
		var cg []cigar.Cigar
		if r.has_cigar != 0 {
			cg = make([]cigar.Cigar, int(r.n_cigar))
			for k := range cg {
				c := allCig[int(r.cigar_off)+k]
				cg[k] = cigar.Cigar{RunLength: int(c.run_length), Op: byte(c.op)}
			}
		}
		seq := reads[i].Seq
		if r.seq_is_rc != 0 {
			seq = reads[i].SeqRc
		}
		out = append(out, giraf.Giraf{QName: reads[i].Name, QStart: int(r.q_start), QEnd: int(r.q_end), Flag: uint8(r.flag), PosStrand: r.pos_strand != 0, Path: path,
			Cigar: cg, AlnScore: int(r.aln_score), MapQ: uint8(r.map_q), Seq: seq, Qual: reads[i].Qual,
			Notes: []giraf.Note{{Tag: []byte{'X', 'O'}, Type: 'Z', Value: "~"}}}) // toGiraf.go:18-30
		if !out[i].PosStrand {
			fastq.ReverseQualUint8Record(out[i].Qual) // toGiraf.go:68-70 (in place, on the read's own slice, as there)
		}
	}
	return out, -1
}

// RoutineFqToGirafHip: the contract of RoutineFqToGiraf (routines.go:12-25), one worker for the whole stream, batchSize reads per device call.
func RoutineFqToGirafHip(g *HipGraph, scoreMatrix [][]int64, batchSize int, threads int, inputChan <-chan fastq.FastqBig, outputChan chan<- giraf.Giraf, wg *sync.WaitGroup) {
	batch := make([]fastq.FastqBig, 0, batchSize)
	flush := func() {
		out, bad := g.gswBatch(batch, scoreMatrix, threads)
		for _, r := range out {
			outputChan <- r
		}
		if bad >= 0 { // the reads before it have left, as with RoutineFqToGiraf; then the panic of that read
			panic(gswPanicText)
		}
		batch = batch[:0]
	}
	for read := range inputChan {
		batch = append(batch, read)
		if len(batch) == batchSize {
			flush()
		}
	}
	flush()
	wg.Done()
}
