"""Shared helpers for the parity tests: seeded synthetic inputs and golden loaders."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
DATA = os.path.join(GOLDEN, "data")

MATRICES = None


def matrices():
    from gonomics_amd import align
    return {"Default": align.DefaultScoreMatrix, "HoxD55": align.HoxD55ScoreMatrix,
            "MouseRat": align.MouseRatScoreMatrix, "HumanChimpTwo": align.HumanChimpTwoScoreMatrix}


def tables():
    with open(os.path.join(GOLDEN, "align_tables.json")) as fh:
        return json.load(fh)


def mutate(rng, seq, sub=0.05, indel=0.01, geo=0.3, alphabet=4):
    """seq with substitutions and geometric-length indels (deterministic given rng)."""
    out = []
    i = 0
    n = len(seq)
    while i < n:
        r = rng.random()
        if r < indel / 2:  # deletion
            i += int(rng.geometric(geo))
            continue
        if r < indel:  # insertion
            k = int(rng.geometric(geo))
            out.extend(rng.integers(0, alphabet, size=k).tolist())
        b = int(seq[i])
        if rng.random() < sub:
            b = int(rng.integers(0, alphabet))
        out.append(b)
        i += 1
    if not out:
        out = [int(rng.integers(0, alphabet))]
    return np.asarray(out, dtype=np.uint8)


def random_pairs(seed, count, nmin, nmax, mmin, mmax, related=0.7, with_n=True):
    rng = np.random.default_rng(seed)
    alphas, betas = [], []
    alphabet = 5 if with_n else 4
    for _ in range(count):
        n = int(rng.integers(nmin, nmax + 1))
        m = int(rng.integers(mmin, mmax + 1))
        a = rng.integers(0, alphabet, size=n).astype(np.uint8)
        if rng.random() < related:
            b = mutate(rng, a, sub=0.08, indel=0.06, geo=0.4, alphabet=alphabet)
            if len(b) > mmax:
                b = b[:mmax]
            if len(b) < mmin:
                b = np.concatenate([b, rng.integers(0, alphabet, size=mmin - len(b)).astype(np.uint8)])
        else:
            b = rng.integers(0, alphabet, size=m).astype(np.uint8)
        alphas.append(a)
        betas.append(b)
    return alphas, betas


def c2_workload(seed, n_pairs, read_len=150, chunk_len=10000):
    """Config C2: one chunk (0.1 % N), reads sampled at uniform offsets, 1 % subs, 0.2 % indel opens."""
    rng = np.random.default_rng(seed)
    chunk = rng.integers(0, 4, size=chunk_len).astype(np.uint8)
    chunk[rng.random(chunk_len) < 0.001] = 4
    reads = np.zeros((n_pairs, read_len), dtype=np.uint8)
    for k in range(n_pairs):
        off = int(rng.integers(0, chunk_len - read_len - 40))
        r = mutate(rng, chunk[off:off + read_len + 40], sub=0.01, indel=0.002, geo=0.5, alphabet=4)
        if len(r) < read_len:
            r = np.concatenate([r, rng.integers(0, 4, size=read_len - len(r)).astype(np.uint8)])
        reads[k] = r[:read_len]
    return reads, chunk


def c3_reads(seed, n_pairs, ref_len, ref_seed, window=10000, read_len=150):
    """config C3: windows at uniform offsets of the synthetic reference, one 150 b read sampled inside each window
    (1 % substitutions, one geometric(0.5)-length indel in ~26 % of the reads).  Vectorised; returns (reads [P, 150], window starts)."""
    from gonomics_amd import _lib
    rng = np.random.default_rng(seed)
    starts = rng.integers(0, ref_len - window, size=n_pairs).astype(np.int64)
    off = rng.integers(0, window - read_len - 64, size=n_pairs)
    x = np.arange(read_len)[None, :]
    has_indel = rng.random(n_pairs) < 0.26
    pos = rng.integers(10, read_len - 10, size=n_pairs)
    ln = np.minimum(rng.geometric(0.5, size=n_pairs), 32)
    is_del = rng.random(n_pairs) < 0.5
    shift = np.where(has_indel[:, None] & (x >= pos[:, None]), np.where(is_del, ln, -ln)[:, None], 0)
    src = (starts + off)[:, None] + np.clip(x + shift, 0, None)
    reads = np.empty((n_pairs, read_len), dtype=np.uint8)
    for lo in range(0, n_pairs, 65536):
        hi = min(n_pairs, lo + 65536)
        reads[lo:hi] = _lib.synthetic_reference_positions(src[lo:hi].reshape(-1), ref_seed).reshape(hi - lo, read_len)
    ins_mask = has_indel[:, None] & (~is_del)[:, None] & (x >= pos[:, None]) & (x < (pos + ln)[:, None])
    reads = np.where(ins_mask, rng.integers(0, 4, size=reads.shape), reads)
    sub = rng.random(reads.shape) < 0.01
    reads = np.where(sub, rng.integers(0, 4, size=reads.shape), reads).astype(np.uint8)
    return np.ascontiguousarray(reads), starts


def synthetic_reference_positions_torch(pos, seed):
    """gonomics_amd._lib.synthetic_reference_positions on a torch int64 tensor (any device): splitmix64 in wrapping int64 arithmetic,
    logical shifts spelled as arithmetic shift + mask.  Used to generate the 10 M reads of config C3 in seconds instead of minutes."""
    import torch

    def s64(v):
        return v - (1 << 64) if v >= (1 << 63) else v

    def lsr(x, k):
        return (x >> k) & ((1 << (64 - k)) - 1)

    w = pos >> 5
    x = (w ^ s64(seed)) + s64(0x9E3779B97F4A7C15)
    x = (x ^ lsr(x, 30)) * s64(0xBF58476D1CE4E5B9)
    x = (x ^ lsr(x, 27)) * s64(0x94D049BB133111EB)
    x = x ^ lsr(x, 31)
    b = ((x >> (2 * (pos & 31))) & 3).to(torch.uint8)
    b[(pos % 50000000 < 1000) & (pos >= 50000000)] = 4
    return b


def c3_reads_torch(seed, n_pairs, ref_len, ref_seed, window=10000, read_len=150, device="cpu"):
    """c3_reads with the per-base work on a torch device (same distributions, its own random stream).  Returns numpy (reads, starts)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    dev = torch.device(device)

    def randint(lo, hi, shape):
        return torch.randint(lo, hi, shape, generator=g, device=dev, dtype=torch.int64)

    starts = randint(0, ref_len - window, (n_pairs,))
    off = randint(0, window - read_len - 64, (n_pairs,))
    has_indel = torch.rand(n_pairs, generator=g, device=dev) < 0.26
    pos = randint(10, read_len - 10, (n_pairs,))
    u = torch.rand(n_pairs, generator=g, device=dev).clamp_min(1e-12)
    ln = torch.clamp(torch.floor(torch.log2(1.0 / u)).to(torch.int64) + 1, max=32)  # geometric(0.5), capped
    is_del = torch.rand(n_pairs, generator=g, device=dev) < 0.5
    reads = torch.empty((n_pairs, read_len), dtype=torch.uint8, device=dev)
    x = torch.arange(read_len, device=dev)[None, :]
    for lo in range(0, n_pairs, 1 << 18):
        hi = min(n_pairs, lo + (1 << 18))
        sl = slice(lo, hi)
        shift = torch.where(has_indel[sl, None] & (x >= pos[sl, None]), torch.where(is_del[sl], ln[sl], -ln[sl])[:, None], torch.zeros((), dtype=torch.int64, device=dev))
        src = (starts[sl] + off[sl])[:, None] + torch.clamp(x + shift, min=0)
        r = synthetic_reference_positions_torch(src.reshape(-1), ref_seed).reshape(hi - lo, read_len)
        ins_mask = has_indel[sl, None] & (~is_del[sl])[:, None] & (x >= pos[sl, None]) & (x < (pos[sl] + ln[sl])[:, None])
        rnd = torch.randint(0, 4, r.shape, generator=g, device=dev, dtype=torch.uint8)
        r = torch.where(ins_mask, rnd, r)
        sub = torch.rand(r.shape, generator=g, device=dev) < 0.01
        rnd2 = torch.randint(0, 4, r.shape, generator=g, device=dev, dtype=torch.uint8)
        reads[sl] = torch.where(sub, rnd2, r)
    return np.ascontiguousarray(reads.cpu().numpy()), starts.cpu().numpy().astype(np.int64)


def assert_same(res_a, res_b, what=""):
    sa, oa, fa = res_a
    sb, ob, fb = res_b
    assert np.array_equal(sa, sb), "scores differ " + what
    assert np.array_equal(fa, fb), "cigar offsets differ " + what
    assert np.array_equal(oa["run_length"], ob["run_length"]), "cigar run lengths differ " + what
    assert np.array_equal(oa["op"], ob["op"]), "cigar ops differ " + what


# ---- callers' serialisation, used to pin results against the reference's golden FILES ----------------
def read_bed4(path):
    rows = []
    with open(path) as fh:
        for line in fh:
            f = line.rstrip("\n").split("\t")
            if len(f) >= 3:
                rows.append((f[0], int(f[1]), int(f[2])))
    return rows


def cigar_to_beds(aln, first_ins, first_del, chrom):
    """The BED derivation of cmd/cigarToBed/cigarToBed.go:89-129 (test-side restatement; aln = [(run, op)])."""
    ins, dele = [], []
    cur = first_ins - 1
    for k in range(len(aln) - 1):
        if aln[k][1] == 0 and aln[k + 1][1] == 1:
            st = cur + aln[k][0] + 1
            ins.append("%s\t%d\t%d\tins\n" % (chrom, st, st + aln[k + 1][0]))
        if aln[k][1] != 2:
            cur += aln[k][0]
    cur = first_del - 1
    for k in range(len(aln) - 1):
        if aln[k][1] == 0 and aln[k + 1][1] == 1:
            st = cur + aln[k][0]
            dele.append("%s\t%d\t%d\tdel\n" % (chrom, st, st + 1))
        if aln[k][1] != 1:
            cur += aln[k][0]
    return "".join(ins), "".join(dele)


def anchor_cases(idx):
    """(alpha, beta, expected_score, expected_cigar_%v) for out_alignment.<idx>.expected.tsv
    (cmd/globalAlignmentAnchor: regions are BED [start-1, end-1) slices, upper-cased; globalAlignmentAnchor.go:378-381)."""
    from gonomics_amd import dna, fasta
    d = os.path.join(DATA, "globalAlignmentAnchor")
    g1 = fasta.ToMap(fasta.Read(os.path.join(d, "hg38.toy.fa")))
    g2 = fasta.ToMap(fasta.Read(os.path.join(d, "rheMac10.toy.fa")))
    b1 = read_bed4(os.path.join(d, "out_hg38_gap.%d.expected.bed" % idx))
    b2 = read_bed4(os.path.join(d, "out_rheMac10_gap.%d.expected.bed" % idx))
    cases = []
    with open(os.path.join(d, "out_alignment.%d.expected.tsv" % idx)) as fh:
        lines = [ln.rstrip("\n").split("\t") for ln in fh if ln.strip()]
    assert len(lines) == len(b1) == len(b2)
    for ln, r1, r2 in zip(lines, b1, b2):
        assert (ln[0], int(ln[1]), int(ln[2])) == r1 and (ln[4], int(ln[5]), int(ln[6])) == r2
        a = dna.AllToUpper(g1[r1[0]][r1[1] - 1:r1[2] - 1].copy())
        b = dna.AllToUpper(g2[r2[0]][r2[1] - 1:r2[2] - 1].copy())
        cases.append((a, b, int(ln[8]), ln[9]))
    return cases


def fmt_v(route):
    return "[" + " ".join("{%d %d}" % (r, o) for r, o in route) + "]"


def view(alpha, beta, route):
    """align.View on (run, op) tuples (align/view.go:37-60)."""
    from gonomics_amd import dna
    a, b = dna.BasesToString(alpha), dna.BasesToString(beta)
    one, two = [], []
    i = j = 0
    for n, op in route:
        if op == 0:
            one.append(a[i:i + n]); two.append(b[j:j + n]); i += n; j += n
        elif op == 1:
            one.append("-" * n); two.append(b[j:j + n]); j += n
        else:
            one.append(a[i:i + n]); two.append("-" * n); i += n
    return "".join(one) + "\n" + "".join(two) + "\n"


# Appendix B of SURVEY.md: derived (NOT reference-stated) known answers for the checkerboard quirks
QUIRK_CASES = [
    ("affine", "Default", -400, -30, 4, "CTCCGTTGCTGCG", "CTCCGTGCGCG", 277, "7M2D4M", "5M1D2M1D4M"),
    ("affine", "Default", -400, -30, 2, "AGGCTTGGCCACG", "AGGCTGTCACG", 482, "5M2D6M", "4M1D1M1D6M"),
    ("affine", "Default", -400, -30, 2, "GTTCC", "TTC", -300, "2D3M", "3M"),
    ("affine", "Default", -400, -30, 2, "AGAACAAGGGGG", "GACAAAGGGG", 251, "2D10M", "10M"),
    ("affine", "HumanChimpTwo", -600, -150, 2, "ATC", "TAGAGGAGA", -2070, "6I3M", "3M"),
    ("const", "Default", -430, 0, 2, "GGCC", "CC", -660, "2D2M", "2M"),
    ("const", "Default", -430, 0, 2, "CTTCAGTAGTCA", "TCAGAGTTCA", -464, "2D10M", "10M"),
]


# Route assertions.  tools/switch_matrix.sh runs the suites under switches that force another route (GNX_FASTPATH, GNX_CLONG,
# GNX_NO_HFORM set OUTSIDE the test): results must not change, but "which path ran" legitimately does -- then only results are checked.
OUTER_ROUTE_SWITCH = any(k in os.environ for k in ("GNX_FASTPATH", "GNX_CLONG", "GNX_NO_HFORM", "GNX_LAT", "GNX_WIDE", "GNX_MEGA_STRIPS", "GNX_W64")) or os.environ.get("GNX_FP_SMALL") == "0"  # (at import: before any monkeypatch)


def shipped_small_batch_rule():
    """True while the library's own routing of small batches is in force (conftest's `routing` fixture: id "shipped"; GNX_FP_SMALL unset):
    batches of fewer than 3 072 one-block reads then take the general path, so "the fast path ran" is not something a small test batch can assert"""
    return os.environ.get("GNX_FP_SMALL") != "1"


def route_switched():
    return OUTER_ROUTE_SWITCH or shipped_small_batch_rule()


def expect_route(timing, route):
    if OUTER_ROUTE_SWITCH or (route == 1 and shipped_small_batch_rule()):
        return
    if route == 0 and timing["fast_path"] == 3:  # the stored-matrix family: small launches run it in the latency geometry (lat_fill_kernel)
        return
    if route == 2 and timing["fast_path"] == 6:  # the snapshot family: launches of up to three pairs (up to 64 long ones) run it with the whole wave on one pair (affine_long64.hip.h, const_long64.hip.h, farm64.hip.h)
        return
    assert timing["fast_path"] == route, (timing["fast_path"], route)
