// gsw_reads.hip.h -- the graph aligner's read path behind ONE C-ABI call per batch of reads ("next" row N2 of SURVEY 8f).
// Part of libgonomics_align_hip.so; included by gnx_align.hip (one translation unit).
// The per-read driver is the C++ host mirror include/gonomics_genomegraph.hpp (GraphSmithWatermanToGiraf, toGiraf.go:17-72; seeds,
// traversals and routes as the reference builds them, Go's slices modelled) on a pool of host threads -- the reference's `-t`
// worker goroutines (genomeGraph/routines.go:12-65) --, with the seed search and the extension DPs of the whole batch on the device
// (gnx_seed_find_batch, gnx_gsw_extend_batch).  What this file adds is the flat boundary: a graph handle that keeps nodes, edges and
// the seed index (resident on the device between calls), reads in as one concatenated buffer, girafs out as arrays.
#pragma once
#include "gonomics_genomegraph.hpp"

struct gnx_gsw_graph {
    gonomics::genomeGraph::GenomeGraph g;
    std::unique_ptr<gonomics::genomeGraph::SeedIndex> index;
    std::mutex mu; // one batch at a time per graph
};

extern "C" {

int gnx_gsw_graph_create(const uint8_t *node_cat, const int64_t *node_off, int64_t n_nodes, const int32_t *edge_from, const int32_t *edge_to,
                         int64_t n_edges, int seed_len, int seed_step, gnx_gsw_graph **out) {
    g_err[0] = 0;
    if (!node_off || n_nodes < 1 || n_nodes > 0x7fffffff || n_edges < 0 || (n_edges > 0 && (!edge_from || !edge_to)) || !out || seed_step < 1) { set_err("bad argument%s", ""); return GNX_EINVAL; }
    try {
        using namespace gonomics::genomeGraph;
        auto h = std::make_unique<gnx_gsw_graph>();
        for (int64_t k = 0; k < n_nodes; k++) {
            const int64_t lo = node_off[k], hi = node_off[k + 1];
            if (lo < 0 || hi < lo || (hi > lo && !node_cat)) { set_err("bad node offsets at node %s%lld", "", (long long)k); return GNX_EINVAL; }
            for (int64_t x = lo; x < hi; x++) if (node_cat[x] > 4) { set_err("a base >= 5 was found in node %s%lld", "", (long long)k); return GNX_EBASE; }
            h->g.AddNode(Bases(node_cat + lo, node_cat + hi));
        }
        for (int64_t e = 0; e < n_edges; e++) {
            if (edge_from[e] < 0 || edge_from[e] >= n_nodes || edge_to[e] < 0 || edge_to[e] >= n_nodes) { set_err("bad edge %s%lld", "", (long long)e); return GNX_EINVAL; }
            GenomeGraph::AddEdge(h->g.Nodes[(size_t)edge_from[e]].get(), h->g.Nodes[(size_t)edge_to[e]].get());
        }
        h->index = std::make_unique<SeedIndex>(h->g, seed_len, seed_step); // (gnx_seed_index_build on the device + the k-mers across node borders)
        *out = h.release();
        return GNX_OK;
    } catch (const gonomics::genomeGraph::GnxFailure &e) { // a library call inside the driver failed: its own code (device, memory, ...)
        if (!g_err[0]) set_err("%s", e.what());
        return e.rc;
    } catch (const std::bad_alloc &) {
        set_err("host allocation failed%s", "");
        return GNX_ENOMEM;
    } catch (const std::exception &e) {
        if (!g_err[0]) set_err("%s", e.what());
        return GNX_EINVAL;
    }
}

void gnx_gsw_graph_free(gnx_gsw_graph *h) { delete h; }

int gnx_gsw_map_reads(gnx_gsw_graph *h, const uint8_t *read_cat, const int64_t *read_off, int64_t n_reads, int paired, const int64_t *scores,
                      int64_t gap_pen, int threads, gnx_giraf **out_girafs, uint32_t **out_nodes, gnx_cigar **out_cigars) {
    g_err[0] = 0;
    if (!h || !read_off || n_reads < 0 || n_reads > 0x3ffffff0 || !scores || !out_girafs || !out_nodes || !out_cigars || (paired && (n_reads & 1))) { set_err("bad argument%s", ""); return GNX_EINVAL; }
    *out_girafs = nullptr; *out_nodes = nullptr; *out_cigars = nullptr;
    try {
        using namespace gonomics::genomeGraph;
        std::lock_guard<std::mutex> lk(h->mu);
        const int T = gswThreads(threads);
        for (int64_t r = 0; r < n_reads; r++) {
            const int64_t lo = read_off[r], hi = read_off[r + 1];
            if (lo < 0 || hi < lo || (hi > lo && !read_cat)) { set_err("bad read offsets at read %s%lld", "", (long long)r); return GNX_EINVAL; }
        }
        std::vector<FastqBig> reads((size_t)n_reads, FastqBig(std::string(), Bases()));
        std::atomic<int64_t> bad{-1};
        parallelFor((size_t)n_reads, T, [&](size_t r) { // (blocks: a read's buffers belong to the worker that will drive it)
            const int64_t lo = read_off[r], hi = read_off[r + 1];
            for (int64_t x = lo; x < hi; x++) if (read_cat[x] > 4) { // the LOWEST failing read is the one reported, whichever worker finds its own first
                int64_t cur = bad.load();
                while ((cur < 0 || (int64_t)r < cur) && !bad.compare_exchange_weak(cur, (int64_t)r)) {}
                return;
            }
            reads[r] = FastqBig(std::string(), Bases(read_cat + lo, read_cat + hi));
        }, /*blocks=*/true);
        if (bad >= 0) { set_err("a base >= 5 was found in read %s%lld", "", (long long)bad.load()); return GNX_EBASE; }
        // (whose index is on the device is checked by the search itself, inside the library's lock: gnx_seed_find_batch_gen)
        std::vector<Giraf> res = paired ? WrapPairGirafBatch(h->g, reads, *h->index, scores, gap_pen, nullptr, /*markPanics=*/true, threads)
                                        : GswBatchToGiraf(h->g, reads, *h->index, scores, gap_pen, nullptr, /*markPanics=*/true, threads);
        int64_t nn = 0, nc = 0;
        for (const Giraf &g : res) if (!g.Panicked) { nn += (int64_t)g.Nodes.size(); nc += (int64_t)g.Cig.size(); }
        gnx_giraf *og = (gnx_giraf *)calloc((size_t)std::max<int64_t>(n_reads, 1), sizeof(gnx_giraf));
        uint32_t *on = (uint32_t *)malloc((size_t)std::max<int64_t>(nn, 1) * 4);
        gnx_cigar *oc = (gnx_cigar *)calloc((size_t)std::max<int64_t>(nc, 1), sizeof(gnx_cigar));
        if (!og || !on || !oc) { free(og); free(on); free(oc); set_err("host allocation failed%s", ""); return GNX_ENOMEM; }
        nn = 0; nc = 0;
        for (int64_t r = 0; r < n_reads; r++) {
            const Giraf &g = res[(size_t)r];
            og[r].node_off = nn; og[r].cigar_off = nc;
            if (!g.Panicked) { nn += (int64_t)g.Nodes.size(); nc += (int64_t)g.Cig.size(); }
        }
        parallelFor((size_t)n_reads, T, [&](size_t r) {
            Giraf &g = res[r];
            gnx_giraf &o = og[r];
            int64_t nn = o.node_off, nc = o.cigar_off;
            o.panicked = g.Panicked ? 1 : 0;
            if (g.Panicked) return; // (the Go process would have died on this read: getLeftTargetBases, search.go:139)
            o.q_start = g.QStart; o.q_end = g.QEnd; o.t_start = g.TStart; o.t_end = g.TEnd; o.aln_score = g.AlnScore;
            o.pos_strand = g.PosStrand ? 1 : 0; o.flag = g.Flag; o.map_q = g.MapQ; o.has_cigar = g.hasCigar ? 1 : 0;
            o.n_nodes = (int64_t)g.Nodes.size(); o.n_cigar = (int64_t)g.Cig.size();
            o.seq_is_rc = (g.Seq == &reads[(size_t)r].SeqRc) ? 1 : 0;
            for (uint32_t n : g.Nodes) on[nn++] = n;
            for (const Cigar &c : g.Cig) { oc[nc].run_length = c.RunLength; oc[nc].op = c.Op; nc++; }
            { Giraf done = std::move(res[r]); FastqBig gone = std::move(reads[r]); } // (freed by the worker that allocated them)
        }, /*blocks=*/true);
        *out_girafs = og; *out_nodes = on; *out_cigars = oc;
        return GNX_OK;
    } catch (const gonomics::genomeGraph::GnxFailure &e) { // a library call inside the driver failed: its own code (device, memory, ...)
        if (!g_err[0]) set_err("%s", e.what());
        return e.rc;
    } catch (const std::bad_alloc &) {
        set_err("host allocation failed%s", "");
        return GNX_ENOMEM;
    } catch (const std::exception &e) {
        if (!g_err[0]) set_err("%s", e.what());
        return GNX_EINVAL;
    }
}

} // extern "C"
