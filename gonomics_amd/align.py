"""Host-side mirror of the reference's `align` package API for the DP hot path.

Same names, argument meaning and error behaviour as the Go functions, so tests read like the
reference's own (/root/reference/align/affineGap_test.go, view_test.go):

  AffineGap, AffineGap_customizeCheckersize      /root/reference/align/affineGap.go:59,73
  ConstGap, ConstGap_customizeCheckersize        /root/reference/align/constGap.go:13,73
  AffineGap_highMem, AffineGapLocal              /root/reference/align/affineGap_highMem.go:99,105
  ConstGap_highMem                               /root/reference/align/constGap_highMem.go:11
  GoAffineGapLocalEngine, TargetQueryPair        /root/reference/align/affineGap_highMem.go:109-125
  View, PrintCigar                               /root/reference/align/view.go:26-60
  Cigar, ColM/ColI/ColD, the four score matrices /root/reference/align/align.go:12-64

Every alignment call goes through the C ABI into the HIP kernels (gonomics_amd/_lib.py); a single
pair is a batch of one.  Go panics map to exceptions: base >= 5 -> IndexError (index out of range),
empty input to a low-memory function (the Go code never terminates) -> ValueError.
"""
from collections import namedtuple

import numpy as np

from . import _lib
from . import dna

ColM, ColI, ColD = 0, 1, 2
Cigar = namedtuple("Cigar", ["RunLength", "Op"])

DefaultScoreMatrix = [
    [91, -114, -31, -123, -44],
    [-114, 100, -125, -31, -43],
    [-31, -125, 100, -114, -43],
    [-123, -31, -114, 91, -44],
    [-44, -43, -43, -44, -43],
]
HoxD55ScoreMatrix = [
    [91, -114, -31, -123, 0],
    [-114, 100, -125, -31, 0],
    [-31, -125, 100, -114, 0],
    [-123, -31, -114, 91, 0],
    [0, 0, 0, 0, 0],
]
MouseRatScoreMatrix = [row[:] for row in HoxD55ScoreMatrix]
HumanChimpTwoScoreMatrix = [
    [90, -330, -236, -356, -208],
    [-330, 100, -318, -236, -196],
    [-236, -318, 100, -330, -196],
    [-356, -236, -330, 90, -208],
    [-208, -196, -196, -208, -202],
]


def _raise(e):
    if e.code == _lib.GNX_EBASE:
        raise IndexError("runtime error: index out of range (dna.Base >= 5 used as a score-matrix index)") from e
    if e.code == _lib.GNX_EEMPTY:
        raise ValueError("empty sequence: the reference's checkerboard loop never terminates on it") from e
    raise e


def _to_route(ops):
    return [Cigar(int(r), int(o)) for r, o in zip(ops["run_length"], ops["op"])]


def _one(params, alpha, beta):
    try:
        score, ops = _lib.align_pair(params, alpha, beta)
    except _lib.GnxError as e:
        _raise(e)
    return score, _to_route(ops)


def AffineGap(alpha, beta, scores, gapOpen, gapExtend):
    return AffineGap_customizeCheckersize(alpha, beta, scores, gapOpen, gapExtend, 10000, 10000)


def AffineGap_customizeCheckersize(alpha, beta, scores, gapOpen, gapExtend, checkersize_i, checkersize_j):
    return _one(_lib.make_params(_lib.GNX_AFFINE_GAP, scores, gapOpen, gapExtend, checkersize_i, checkersize_j), alpha, beta)


def ConstGap(alpha, beta, scores, gapPen):
    return ConstGap_customizeCheckersize(alpha, beta, scores, gapPen, 10000, 10000)


def ConstGap_customizeCheckersize(alpha, beta, scores, gapPen, checkersize_i, checkersize_j):
    return _one(_lib.make_params(_lib.GNX_CONST_GAP, scores, gapPen, 0, checkersize_i, checkersize_j), alpha, beta)


def AffineGap_highMem(alpha, beta, scores, gapOpen, gapExtend):
    return _one(_lib.make_params(_lib.GNX_AFFINE_GAP_HIGHMEM, scores, gapOpen, gapExtend), alpha, beta)


def AffineGapLocal(target, query, scores, gapOpen, gapExtend):
    return _one(_lib.make_params(_lib.GNX_AFFINE_GAP_LOCAL, scores, gapOpen, gapExtend), target, query)


def ConstGap_highMem(alpha, beta, scores, gapPen):
    return _one(_lib.make_params(_lib.GNX_CONST_GAP_HIGHMEM, scores, gapPen), alpha, beta)


def AlignBatch(params, alphas, betas):
    """Batched form used by loops over independent pairs (cmd/globalAlignmentAnchor.go:352-384).
    Returns [(score, route), ...] in input order."""
    try:
        scores, ops, off = _lib.align_batch(params, alphas, betas)
    except _lib.GnxError as e:
        _raise(e)
    return [(int(scores[k]), _to_route(ops[off[k]:off[k + 1]])) for k in range(len(alphas))]


def AffineGapChunk(alpha, beta, scores, gapOpen, gapExtend, chunkSize):
    """align.AffineGapChunk (/root/reference/align/affineGap_highMem.go:227-268)."""
    try:
        sc, ops, off = _lib.affine_gap_chunk_batch(_lib.make_params(_lib.GNX_AFFINE_GAP_HIGHMEM, scores, gapOpen, gapExtend), chunkSize, [alpha], [beta])
    except _lib.GnxError as e:
        _raise(e)
    return int(sc[0]), _to_route(ops[off[0]:off[1]])


def _groups_to_blocks(groups):
    return [np.stack([np.asarray(f.Seq, dtype=np.uint8) for f in g]) for g in groups]


def multipleAffineGapBatch(groups, pairs, scores, gapOpen, gapExtend, chunkSize=1):
    """multipleAffineGap / multipleAffineGapChunk (affineGap_highMem.go:270-353) for many pairs of fasta groups at once."""
    try:
        sc, ops, off = _lib.multiple_affine_gap_batch(_lib.make_params(_lib.GNX_AFFINE_GAP_HIGHMEM, scores, gapOpen, gapExtend), chunkSize,
                                                      _groups_to_blocks(groups), pairs)
    except _lib.GnxError as e:
        _raise(e)
    return [(int(sc[k]), _to_route(ops[off[k]:off[k + 1]])) for k in range(len(pairs))]


def mergeMultipleAlignments(alpha, beta, route):
    """align/multiAlign.go:112-153: merge two fasta groups along a cigar (host-side, no DP)."""
    from .fasta import Fasta
    total = sum(c.RunLength for c in route)
    rows = [np.full(total, dna.Gap, dtype=np.uint8) for _ in range(len(alpha) + len(beta))]
    acol = bcol = col = 0
    for c in route:
        n = c.RunLength
        if c.Op in (ColM, ColD):
            for k, f in enumerate(alpha):
                rows[k][col:col + n] = np.asarray(f.Seq, dtype=np.uint8)[acol:acol + n]
        if c.Op in (ColM, ColI):
            for k, f in enumerate(beta):
                rows[len(alpha) + k][col:col + n] = np.asarray(f.Seq, dtype=np.uint8)[bcol:bcol + n]
        if c.Op != ColI:
            acol += n
        if c.Op != ColD:
            bcol += n
        col += n
    return [Fasta(f.Name, rows[k]) for k, f in enumerate(list(alpha) + list(beta))]


def _all_seq(records, scoreMatrix, gapOpen, gapExtend, chunkSize, batch_fn):
    # multiAlign.go:27-78: progressive alignment, merging the best-scoring pair of groups each round
    # (first strict maximum in x<y order, nearestGroups :27-41); one batched call per round.
    groups = [[r] for r in records]
    while len(groups) > 1:
        pairs = [(x, y) for x in range(len(groups) - 1) for y in range(x + 1, len(groups))]
        res = batch_fn(groups, pairs, scoreMatrix, gapOpen, gapExtend, chunkSize)
        best, best_score = None, None
        for (x, y), (score, route) in zip(pairs, res):
            if best_score is None or score > best_score:
                best, best_score = (x, y, route), score
        x, y, route = best
        groups[x] = mergeMultipleAlignments(groups[x], groups[y], route)
        groups[y] = groups[-1]
        groups = groups[:-1]
    return groups[0]


def AllSeqAffine(records, scoreMatrix, gapOpen, gapExtend):
    """align.AllSeqAffine (/root/reference/align/multiAlign.go:59-66)."""
    return _all_seq(records, scoreMatrix, gapOpen, gapExtend, 1, multipleAffineGapBatch)


def AllSeqAffineChunk(records, scoreMatrix, gapOpen, gapExtend, chunkSize):
    """align.AllSeqAffineChunk (/root/reference/align/multiAlign.go:70-78)."""
    return _all_seq(records, scoreMatrix, gapOpen, gapExtend, chunkSize, multipleAffineGapBatch)


class TargetQueryPair:
    """align.TargetQueryPair (affineGap_highMem.go:110-115)."""
    __slots__ = ("Target", "Query", "Score", "Cigar")

    def __init__(self, Target=None, Query=None, Score=0, Cigar=None):
        self.Target, self.Query, self.Score, self.Cigar = Target, Query, Score, Cigar


class _Engine:
    """FIFO batched stand-in for the two channels of GoAffineGapLocalEngine: `send` queues a pair,
    `recv` returns results strictly in input order (the Go engine has one worker goroutine)."""

    def __init__(self, scores, gapOpen, gapExtend, max_batch=1000):
        self._params = _lib.make_params(_lib.GNX_AFFINE_GAP_LOCAL, scores, gapOpen, gapExtend)
        self._pending, self._done, self._max = [], [], max_batch
        self._closed = False

    def send(self, pair):
        if self._closed:
            raise RuntimeError("send on closed channel")
        self._pending.append(pair)
        if len(self._pending) >= self._max:
            self._flush()

    def _flush(self):
        if not self._pending:
            return
        res = AlignBatch(self._params, [p.Target for p in self._pending], [p.Query for p in self._pending])
        for p, (s, c) in zip(self._pending, res):
            p.Score, p.Cigar = s, c
            self._done.append(p)
        self._pending = []

    def recv(self):
        if not self._done:
            self._flush()
        if not self._done:
            if self._closed:
                return None
            raise RuntimeError("recv would block: nothing was sent")
        return self._done.pop(0)

    def close(self):
        self._closed = True

    def __iter__(self):
        while True:
            r = self.recv() if (self._done or self._pending) else None
            if r is None:
                return
            yield r


def GoAffineGapLocalEngine(scores, gapOpen, gapExtend):
    """Returns (inputs, outputs); both are the same FIFO engine object (inputs.send / outputs.recv)."""
    e = _Engine(scores, gapOpen, gapExtend)
    return e, e


def _col_rune(op):
    if op == ColM:
        return "M"
    if op == ColI:
        return "I"
    if op == ColD:
        return "D"
    raise ValueError("Error: unexpected value when converting colType to rune %d" % op)


def PrintCigar(operations):
    return "".join("%d%s" % (c.RunLength, _col_rune(c.Op)) for c in operations)


def FormatCigar(operations):
    """Go's fmt %v of a []Cigar, e.g. `[{2 0} {1 1}]` (cmd/globalAlignmentAnchor.go:23-27)."""
    return "[" + " ".join("{%d %d}" % (c.RunLength, c.Op) for c in operations) + "]"


def View(alpha, beta, operations):
    one, two = [], []
    i = j = 0
    a = dna.BasesToString(np.asarray(alpha, dtype=np.uint8))
    b = dna.BasesToString(np.asarray(beta, dtype=np.uint8))
    for c in operations:
        n = c.RunLength
        if c.Op == ColM:
            one.append(a[i:i + n]); two.append(b[j:j + n]); i += n; j += n
        elif c.Op == ColI:
            one.append("-" * n); two.append(b[j:j + n]); j += n
        elif c.Op == ColD:
            one.append(a[i:i + n]); two.append("-" * n); i += n
    return "".join(one) + "\n" + "".join(two) + "\n"
