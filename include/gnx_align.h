/*
 * gnx_align.h -- C ABI of libgonomics_align_hip.so: the MI355X (gfx950) drop-in for the pairwise DP hot
 * path of gonomics' `align` package.
 *
 * The reference (vertgenlab/gonomics) is pure Go and has no FFI layer.  The boundary it offers for this
 * path is the exported Go API of package align; each entry point below names the Go function(s) a cgo
 * shim would route through it (see INTEGRATION.md for the shim):
 *
 *   align.AffineGap(alpha, beta []dna.Base, scores [][]int64, gapOpen, gapExtend int64) (int64, []Cigar)
 *                                                              /root/reference/align/affineGap.go:59
 *   align.AffineGap_customizeCheckersize(..., checkersize_i, checkersize_j int)        affineGap.go:73
 *   align.ConstGap(alpha, beta, scores, gapPen) (int64, []Cigar)                        constGap.go:13
 *   align.ConstGap_customizeCheckersize(..., checkersize_i, checkersize_j int)         constGap.go:73
 *   align.AffineGap_highMem / align.AffineGapLocal(target, query, ...)      affineGap_highMem.go:99,105
 *   align.ConstGap_highMem                                                      constGap_highMem.go:11
 *   align.GoAffineGapLocalEngine (batched, FIFO)                              affineGap_highMem.go:120
 *
 * Conventions
 *   - bases are dna.Base bytes (A=0 C=1 G=2 T=3 N=4; /root/reference/dna/dna.go:5-21); any byte >= 5
 *     makes the Go code panic (index out of range on the 5x5 matrix) -> here: GNX_EBASE.
 *   - `scores` is the [][]int64 matrix flattened row-major: scores[alphaBase*5 + betaBase].
 *   - results are bit-exact with the reference: score (int64) and the run-length CIGAR in alignment
 *     order, including the low-memory checkerboard traceback quirks for n or m > checkersize.
 *   - gnx_cigar has the memory layout of Go's align.Cigar{RunLength int64; Op ColType(uint8)} on amd64.
 *   - inputs are borrowed and never written; outputs returned through gnx_cigar** / int64_t** are
 *     malloc'd by the library and released with gnx_free().
 *   - all functions return GNX_OK (0) or a GNX_E* code; gnx_last_error() gives the text.
 *   - there is NO CPU fallback: without a usable HIP device every compute entry returns GNX_EDEVICE.
 */
#ifndef GNX_ALIGN_H
#define GNX_ALIGN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GNX_OK 0
#define GNX_EINVAL 1    /* bad argument (null pointer, negative size, unknown mode) */
#define GNX_EBASE 2     /* a base >= 5: the Go code would panic */
#define GNX_EEMPTY 3    /* empty sequence in a low-memory mode: the Go code would never terminate */
#define GNX_ERANGE 4    /* lengths x penalties exceed the int32 range of a mode that keeps absolute keys (the chunk / graph-extension variants).  The align functions
                         * proper (AffineGap*, ConstGap*, AffineGapLocal) have no range limit: what bounds them is (a) 2^30 - 1 bases per sequence -- longer: GNX_EINVAL --
                         * and (b) device memory: a pair whose bottom rows + snapshots fit neither the snapshot path nor row panels, or whose stored matrix (the int64
                         * kernel: AffineGapLocal, gapOpen > 0 beyond int32) does not fit, returns GNX_ENOMEM, not GNX_ERANGE.  A single oversize pair is given what the
                         * device has free, beyond the workspace limit of gnx_init (it cannot run any other way). */
#define GNX_EDEVICE 5   /* no HIP device / HIP runtime error */
#define GNX_ENOMEM 6    /* host or device allocation failed / workspace too small for one pair */
#define GNX_ECAPACITY 7 /* caller-provided device CIGAR buffer too small (device entry point) */
#define GNX_ETRACE 8    /* impossible traceback value: the Go code would log.Fatalf */
#define GNX_EDIVZERO 9  /* scoreColumnMatch over a column pair of gaps only: the Go code panics (integer divide by zero, multiAlign.go:101) */
#define GNX_ESTALE 10   /* gnx_seed_find_batch_gen: the resident seed index is not the generation named (set it again) */

/* align.ColType, /root/reference/align/align.go:12-18 */
#define GNX_COL_M 0
#define GNX_COL_I 1
#define GNX_COL_D 2

/* align.Cigar, /root/reference/align/align.go:21-24 */
typedef struct gnx_cigar {
    int64_t run_length;
    uint8_t op;
    uint8_t _pad[7];
} gnx_cigar;

typedef enum gnx_mode {
    GNX_AFFINE_GAP = 0,         /* AffineGap / AffineGap_customizeCheckersize (low-memory checkerboard semantics) */
    GNX_CONST_GAP = 1,          /* ConstGap / ConstGap_customizeCheckersize */
    GNX_AFFINE_GAP_HIGHMEM = 2, /* AffineGap_highMem */
    GNX_AFFINE_GAP_LOCAL = 3,   /* AffineGapLocal(target=alpha, query=beta), GoAffineGapLocalEngine */
    GNX_CONST_GAP_HIGHMEM = 4   /* ConstGap_highMem */
} gnx_mode;

typedef struct gnx_params {
    int32_t mode;          /* gnx_mode */
    int32_t _reserved;
    int64_t scores[25];    /* scores[a*5+b] */
    int64_t gap_open;      /* affine: gapOpen; const: gapPen */
    int64_t gap_extend;    /* affine: gapExtend; const: ignored */
    int64_t checkersize_i; /* low-memory modes only; 10000 for AffineGap / ConstGap */
    int64_t checkersize_j;
} gnx_params;

/* Per-call kernel timings of the most recent compute call of the process (context 0; with several contexts: the slowest
 * context's times, cells and bytes summed) -- HIP events on the stream the kernels ran on.  cells = sum over pairs of n*m. */
typedef struct gnx_timing {
    double fill_ms;      /* DP fill kernel(s): the dominant kernel */
    double traceback_ms; /* traceback count + scan + write kernels */
    double total_ms;     /* first launch to last kernel end */
    int64_t cells;
    int64_t n_launches;  /* number of fill launches (sub-batches) */
    int64_t trace_bytes; /* direction-matrix / checkpoint bytes written by the fill kernels */
    double dominant_ms;  /* summed duration of the dominant kernel's launches (fast path: the forward sweep; general
                            path: the fill kernel) -- what roofline.achieved is computed from */
    int64_t dominant_launches;
    int32_t fast_path;   /* 1: the affine fast path ran (short alpha x long beta; AffineGapLocal: short query x long target);
                            2: the constant-gap path without a stored direction matrix (const_long.hip.h); 0: full direction matrix */
    int32_t _pad;
    /* host-buffer entry points only (wall clock inside the library): */
    double host_ms;      /* entry to return of the whole call */
    double stage0_ms;    /* exposed upload of the first sub-batch (later ones run under the kernels) */
    double fetch_ms;     /* gather on device 0 + D2H of scores / offsets / CIGAR runs */
    /* multi-context calls (gnx_init_devices): */
    int32_t transport;   /* what carried the last broadcast / gather: 0 one context, 1 RCCL over xGMI, 2 peer copies (GNX_RCCL=0, or a
                            device listed twice), 3 peer copies AFTER a RCCL call failed (the text is in gnx_last_error) */
    int32_t n_contexts;  /* contexts the call was sharded over */
    double gather_ms;    /* the gather on device 0 alone (part of fetch_ms) */
    double bcast_ms;     /* broadcast of the shared beta buffer of this call (0 with a resident reference) */
} gnx_timing;

/* ---- lifecycle ---------------------------------------------------------------------------------- */
int gnx_device_count(void);
/* Bind this process to HIP device `device` (one process per GPU).  workspace_bytes = upper bound for the
 * library's device scratch (direction matrices etc.); 0 picks a default from free memory. */
int gnx_init(int device, int64_t workspace_bytes);
/* One context per GPU inside ONE host process (what a Go program that calls align.* needs in order to use a whole node:
 * the reference's own parallel drivers are worker pools inside one process, /root/reference/genomeGraph/routines.go:12-65).
 * devices = NULL: devices 0 .. n_devices-1; n_devices <= 0: every visible GPU.  Afterwards the host-buffer entry points below cut
 * each batch into contiguous blocks of equal DP cells, one block per context (one worker thread each), upload a shared beta
 * buffer / the resident reference once and broadcast it with RCCL over xGMI, and gather scores / offsets / CIGARs on device 0
 * (grouped ncclSend / ncclRecv) in input order.  RCCL (librccl.so) is loaded with dlopen here, never for one GPU.
 * A device may be listed more than once (flow tests on a 1-GPU box): such contexts exchange by plain copies instead of RCCL.
 * Environment: GNX_RCCL=0 peer copies instead of RCCL; GNX_RCCL=1 build the communicator even for a single device. */
int gnx_init_devices(int n_devices, const int *devices, int64_t workspace_bytes_per_device);
int gnx_n_devices(void);
void gnx_shutdown(void);
/* Text of the calling thread's last error; if that thread has none, the most recent error of the process (cgo may run the failing
 * call and this one on different OS threads unless the shim pins the goroutine). */
const char *gnx_last_error(void);
void gnx_free(void *p);

/* ---- resident reference (configs C3 / C4: reads against windows of one genome) ------------------ */
/* Upload `ref` (dna.Base bytes) once; it stays on the device(s) until the next call or gnx_shutdown -- PACKED: 2 bits per base plus a
 * sparse list of the 64-base blocks that hold anything but A C G T (N; bytes >= 5, which make GNX_EBASE only when an alignment's
 * window touches them, like the Go code's panic): 4.4e9 bases take 1.1 GB of HBM and of RCCL broadcast instead of 4.4 GB.  It is
 * packed on device 0 (256 MB of bases at a time) and, with several contexts, broadcast over RCCL; the sweep kernels read the packed
 * words directly (len = 0 releases it).  Replaces passing the same target slice to every align.* call of a loop
 * (/root/reference/cmd/globalAlignmentAnchor/globalAlignmentAnchor.go:352-384 re-reads its two genomes per anchor). */
int gnx_set_reference(const uint8_t *ref, int64_t len);
/* What is resident: bases, bytes of device memory per context, 64-base blocks on the exception list. */
int gnx_reference_info(int64_t *out_bases, int64_t *out_device_bytes, int64_t *out_exception_blocks);
/* Synthetic reference of SURVEY 8d (config C3), generated on the device: base(pos) = 2 bits of splitmix64(seed ^ (pos / 32)) at
 * bit 2*(pos % 32), with an N run of 1000 bases at every multiple of 5e7 except 0.  For benchmarks and tests (3e9 bases do not
 * have to cross PCIe); nothing in the reference corresponds to it. */
int gnx_set_reference_synthetic(int64_t len, uint64_t seed);
/* Batch of reads (alpha_cat / alpha_off as in gnx_align_batch) against windows (ref_start[p], ref_len[p]) of the resident
 * reference: beta = reference[ref_start[p] .. ref_start[p] + ref_len[p]).  Only the reads and 16 bytes of window table per pair
 * cross PCIe.  Large batches are cut into sub-batches whose uploads run under the kernels of the sub-batch before (pinned
 * staging, second stream); results accumulate on the device and come back in one transfer into pinned memory (gnx_free). */
int gnx_align_batch_by_offset(const gnx_params *p, int64_t n_pairs, const uint8_t *alpha_cat, const int64_t *alpha_off,
                              const int64_t *ref_start, const int64_t *ref_len,
                              int64_t *out_score, gnx_cigar **out_ops, int64_t **out_ops_off);

/* ---- host-buffer entry points (what the cgo shim binds) ------------------------------------------ */
/* Batch of independent pairs.  Pair p is alpha_cat[alpha_off[p] .. alpha_off[p+1]) vs
 * beta_cat[beta_off[p] .. beta_off[p+1]).  Outputs: out_score[n_pairs]; *out_ops = concatenated CIGARs,
 * (*out_ops_off)[n_pairs+1] their boundaries.  Replaces a serial loop of align.AffineGap* / ConstGap*
 * calls (cmd/globalAlignmentAnchor/globalAlignmentAnchor.go:352-384) and the FIFO engine
 * (affineGap_highMem.go:120-179): results are in input order. */
int gnx_align_batch(const gnx_params *p, int64_t n_pairs,
                    const uint8_t *alpha_cat, const int64_t *alpha_off,
                    const uint8_t *beta_cat, const int64_t *beta_off,
                    int64_t *out_score, gnx_cigar **out_ops, int64_t **out_ops_off);

/* Same, but each sequence is a window (start,len) into a shared buffer, so many pairs may reference one
 * resident chunk / genome (faChunkAlign-style read-vs-chunk batches). */
int gnx_align_batch_windows(const gnx_params *p, int64_t n_pairs,
                            const uint8_t *alpha_buf, int64_t alpha_buf_len, const int64_t *alpha_start, const int64_t *alpha_len,
                            const uint8_t *beta_buf, int64_t beta_buf_len, const int64_t *beta_start, const int64_t *beta_len,
                            int64_t *out_score, gnx_cigar **out_ops, int64_t **out_ops_off);

/* One pair == batch of 1: the body of the Go-signature functions.  Safe and EFFICIENT from many threads at once (a pool of
 * goroutines calling align.* is the reference's house pattern, /root/reference/genomeGraph/routines.go:12-65): concurrent calls with
 * equal parameters are combined into one device batch by whichever caller holds the library's lock; each caller still gets the
 * result, return code and error text of its own pair. */
int gnx_align_pair(const gnx_params *p, const uint8_t *alpha, int64_t n, const uint8_t *beta, int64_t m,
                   int64_t *out_score, gnx_cigar **out_ops, int64_t *out_n_ops);

/* ---- device-resident entry point (bench.py, multi-GPU shards) ------------------------------------ */
/* All d_* pointers are device memory on the bound device; h_* are host copies of the window tables used
 * for planning.  Kernels are enqueued on `stream` (a hipStream_t, may be NULL) and the call returns
 * after they finished.  d_ops has room for ops_capacity elements; *out_total_ops receives the number
 * used (GNX_ECAPACITY if it did not fit). */
int gnx_align_batch_device(const gnx_params *p, int64_t n_pairs,
                           const uint8_t *d_alpha_buf, const int64_t *d_alpha_start, const int64_t *d_alpha_len,
                           const uint8_t *d_beta_buf, const int64_t *d_beta_start, const int64_t *d_beta_len,
                           const int64_t *h_alpha_len, const int64_t *h_beta_len,
                           int64_t *d_score, gnx_cigar *d_ops, int64_t ops_capacity, int64_t *d_ops_off,
                           int64_t *out_total_ops, void *stream);

int gnx_get_timing(gnx_timing *out);
/* Diagnostics (tests only; nothing in the reference corresponds to it): launch n_workgroups workgroups that each hold one CU's whole
 * LDS and spin for `milliseconds` on a stream of their own, and return at once -- the pipelined launches of the library must make
 * progress whatever else occupies the device (no workgroup waits for work that has not been claimed by a running one). */
int gnx_debug_occupy(int n_workgroups, int milliseconds); /* refused with GNX_EINVAL unless the process runs with GNX_DEBUG_ENTRY=1 */
/* Diagnostics: which = 0: number of items of pipelined launches (strips, row-block levels) that were run by a workgroup other
 * than their own since the last reset -- the abnormal path of the claim protocol, which tests/test_ticket.py forces and then
 * proves to have run.  which = 1 / 2: combined batches run for concurrent gnx_align_pair calls / pairs served by them.
 * which = 3 / 4: checkerboard edges the affine walks of the snapshot path crossed upwards where the restart rule of
 * /root/reference/align/affineGap.go:305 (quirk Q1) changed the state / all such crossings -- each of the former can cost the CIGAR
 * at most one gap open against the score, which is what the tests of megabase pairs (no oracle finishes them) check.
 * which = 5 / 6: rows per lane / snapshot spacing of the last 64-lane affine sweep (bench.py prices its design bytes with them).
 * reset != 0 zeroes the counter after reading (3 and 4 together). */
int gnx_debug_counter(int which, int reset, int64_t *out);

/* ---- "next" row N1: chunk and multiple-alignment variants (what cmd/faChunkAlign and popgen/dunn.go run) ---- */
/* align.AffineGapChunk (/root/reference/align/affineGap_highMem.go:227-268): the affine DP over chunks of chunk_size
 * bases (cell score = ungapped score of two chunks, gapExtend*chunkSize, run lengths in bases).  p->mode must be
 * GNX_AFFINE_GAP_HIGHMEM.  Lengths that are not multiples of chunk_size -> GNX_EINVAL (the Go code log.Fatalf's). */
int gnx_affine_gap_chunk_batch(const gnx_params *p, int64_t chunk_size, int64_t n_pairs,
                               const uint8_t *alpha_cat, const int64_t *alpha_off, const uint8_t *beta_cat, const int64_t *beta_off,
                               int64_t *out_score, gnx_cigar **out_ops, int64_t **out_ops_off);

/* align.multipleAffineGap (chunk_size 1, affineGap_highMem.go:270-306) and multipleAffineGapChunk (:308-353) on a batch
 * of pairs of alignment blocks ("groups"): group g holds group_nseq[g] sequences of group_len[g] bases, sequence-major,
 * at group_bases[group_off[g] ..); bases may be lower case or dna.Gap.  Pair q aligns group pair_a[q] against pair_b[q]
 * with the column score of align/multiAlign.go:82-110.  This is the inner loop of AllSeqAffine / AllSeqAffineChunk
 * (multiAlign.go:27-78): one call evaluates all x<y group pairs of a progressive-alignment round. */
int gnx_multiple_affine_gap_batch(const gnx_params *p, int64_t chunk_size, int64_t n_groups, const uint8_t *group_bases,
                                  const int64_t *group_off, const int32_t *group_nseq, const int64_t *group_len,
                                  int64_t n_pairs, const int32_t *pair_a, const int32_t *pair_b,
                                  int64_t *out_score, gnx_cigar **out_ops, int64_t **out_ops_off);

/* ---- "next" row N2: the seed-extension DPs of the graph aligner (what cmd/gsw spends its DP time in) ------------------ */
/* genomeGraph.LeftDynamicAln  (/root/reference/genomeGraph/search.go:234-276): constant gap, borders 0, cell values clamped
 *   at 0; traceback from (len(alpha), len(beta)) while the cell value is > 0.  Returns m[n][m], the route and the (i, j) where
 *   the walk stopped.
 * genomeGraph.RightDynamicAln (search.go:278-321): constant gap with Needleman-Wunsch borders; the alignment ends at the first
 *   row-major cell holding the maximum (> 0) and is traced back to (0, 0).  Returns that maximum, the route and (maxI, maxJ).
 * Both use cigar.TripleMaxTrace (cigar/tools.go:58-66: M >= I >= D).  The route is in TRACEBACK order, exactly as the
 * reference builds it when it is called with an empty dynamicScore.route (callers reverse it, search.go:196,228); ops are
 * GNX_COL_M/I/D for cigar 'M'/'I'/'D', gnx_cigar has the layout of cigar.Cigar{RunLength int; Op byte} on amd64.
 * gap_pen must be <= 0 (gsw passes -600, search.go:176,210).  RIGHT packs (score, column) into one int32 per row:
 * sequences up to 4095 bases and (n+m+2)*max|penalty| < 2^19, else GNX_ERANGE (the reference's matrix is 2480 x 2480,
 * search.go:22).  One call replaces the per-seed calls of GraphSmithWatermanToGiraf (toGiraf.go:58-59) for a batch of reads. */
#define GNX_GSW_LEFT 0
#define GNX_GSW_RIGHT 1
int gnx_gsw_extend_batch(int side, const int64_t *scores, int64_t gap_pen, int64_t n_pairs,
                         const uint8_t *alpha_cat, const int64_t *alpha_off, const uint8_t *beta_cat, const int64_t *beta_off,
                         int64_t *out_score, int64_t *out_end_i, int64_t *out_end_j, gnx_cigar **out_ops, int64_t **out_ops_off);

/* The graph aligner's read path, one call per batch of reads: genomeGraph.GraphSmithWatermanToGiraf (/root/reference/genomeGraph/toGiraf.go:17-72)
 * for every read -- seeds (seedMapMemPool, search.go:549-590), seedCouldBeBetter pruning (index.go:102-121), Left / RightAlignTraversal
 * (search.go:166-232) with their DPs -- and, with paired != 0, WrapPairGiraf's flags (toGiraf.go:117-140; reads 2k / 2k+1 = the mates of pair k).
 * What the reference spreads over `-t` worker goroutines (genomeGraph/routines.go:12-65, cmd/gsw) runs here as: seed search and extension
 * DPs of the whole batch on the device, the per-read bookkeeping between them on `threads` host threads (0 = GNX_GSW_THREADS, else
 * min(hardware threads, 16)).  The graph handle keeps the nodes, the edges and the seed index (IndexGenomeIntoMap, index.go:21-59; resident on
 * the device between calls).  Results in input order: out_girafs[r] with its node ids at out_nodes[node_off ..) and its cigar
 * (cigar.Cigar{RunLength, Op}: op = the ASCII byte 'M' 'I' 'D' 'S', as cigar.Cigar.Op holds it) at out_cigars[cigar_off ..); a read on
 * which the Go code panics (getLeftTargetBases with a short Prev node, search.go:139) has panicked = 1 and nothing else.  The three
 * arrays are malloc'd: gnx_free().  Not pinned by the reference's tests (parity contract: DESIGN.md section 6). */
typedef struct gnx_gsw_graph gnx_gsw_graph;
typedef struct gnx_giraf {
    int64_t q_start, q_end, t_start, t_end, aln_score; /* giraf.Giraf: QStart, QEnd, Path.TStart, Path.TEnd, AlnScore */
    int64_t node_off, n_nodes;                         /* Path.Nodes */
    int64_t cigar_off, n_cigar;                        /* Cigar (has_cigar = 0: nil) */
    int32_t pos_strand, flag, map_q, has_cigar;
    int32_t seq_is_rc;                                 /* Seq is the read's reverse complement (1) or the read (0) */
    int32_t panicked;
} gnx_giraf;
int gnx_gsw_graph_create(const uint8_t *node_cat, const int64_t *node_off, int64_t n_nodes, const int32_t *edge_from, const int32_t *edge_to,
                         int64_t n_edges, int seed_len, int seed_step, gnx_gsw_graph **out);
void gnx_gsw_graph_free(gnx_gsw_graph *g);
int gnx_gsw_map_reads(gnx_gsw_graph *g, const uint8_t *read_cat, const int64_t *read_off, int64_t n_reads, int paired, const int64_t *scores,
                      int64_t gap_pen, int threads, gnx_giraf **out_girafs, uint32_t **out_nodes, gnx_cigar **out_cigars);

/* ---- "next" row N4: the seed index and the seed search of the graph aligner (what cmd/gsw does before its DPs) ----------------- */
/* genomeGraph.IndexGenomeIntoMap (/root/reference/genomeGraph/index.go:21-43) for the k-mers inside nodes: node k is
 * node_cat[node_off[k] .. node_off[k+1]) (dna.Base bytes, N allowed); positions 0, seed_step, ... of every node; k-mers with an N
 * are skipped.  Output = the map as two arrays sorted by key (k-mer code, index.go `dnaToNumber`), equal keys in the reference's
 * insertion order (node, position); locations are `node << 32 | pos` (ChromAndPosToNumber).  malloc'd, gnx_free().  The k-mers
 * that run across node borders (index.go:34-38) are left to the host, which merges them and calls gnx_seed_index_set. */
int gnx_seed_index_build(const uint8_t *node_cat, const int64_t *node_off, int64_t n_nodes, int seed_len, int seed_step,
                         uint64_t **out_keys, uint64_t **out_locs, int64_t *out_n);
/* Make an index (+ the nodes, packed like dnaTwoBit does it) resident on the device for gnx_seed_find_batch. */
int gnx_seed_index_set(const uint64_t *keys, const uint64_t *locs, int64_t n_index, const uint8_t *node_cat, const int64_t *node_off,
                       int64_t n_nodes, int seed_len);
/* The hash-lookup / exact-match part of genomeGraph.seedMapMemPool (search.go:549-590) for a batch of reads: for every read
 * position and strand (0 = read, 1 = reverse complement) every index hit, extended to the left inside its node
 * (dnaTwoBit.CountLeftMatches) and from there to the right (CountRightMatches, the first step of extendToTheRightDev).  Hits of
 * read r are out_hits[out_hit_off[r] .. out_hit_off[r+1]) in the reference's order of discovery.  A hit whose right end reaches
 * the end of its node continues into the Next nodes on the host (as does the leftward continuation over Prev edges). */
typedef struct gnx_seed_hit {
    int32_t read_start; /* the read position whose k-mer hit */
    int32_t strand;
    int32_t node, node_start; /* target of the extended seed part */
    int32_t q_start;          /* its start in the read (strand 1: in the reverse complement) */
    int32_t right;            /* its length */
} gnx_seed_hit;
int gnx_seed_find_batch(const uint8_t *read_cat, const int64_t *read_off, int64_t n_reads, gnx_seed_hit **out_hits, int64_t **out_hit_off);
/* The device holds ONE resident index; gnx_seed_index_set + gnx_seed_find_batch are two calls, so another thread's set can land between
 * them.  Callers that share the process with other users of the index (gnx_gsw_map_reads does) use this pair instead: the set returns
 * the GENERATION it installed, the search names it and is refused with GNX_ESTALE -- inside the lock that runs the search -- when the
 * resident index is no longer that one; the caller then sets again.  (The plain pair above stays for single-threaded callers.) */
int gnx_seed_index_set_gen(const uint64_t *keys, const uint64_t *locs, int64_t n_index, const uint8_t *node_cat, const int64_t *node_off,
                           int64_t n_nodes, int seed_len, uint64_t *out_generation);
int gnx_seed_find_batch_gen(uint64_t generation, const uint8_t *read_cat, const int64_t *read_off, int64_t n_reads, gnx_seed_hit **out_hits, int64_t **out_hit_off);

#ifdef __cplusplus
}
#endif
#endif /* GNX_ALIGN_H */
