"""Test-side progressive alignment with the CPU oracle as the pairwise engine (mirrors align/multiAlign.go:27-78),
used to pin the oracle's N1 functions against the reference's multi-fasta goldens."""
import numpy as np

import oracle
from gonomics_amd import dna
from gonomics_amd.fasta import Fasta


def merge(alpha, beta, route):
    total = sum(r for r, _ in route)
    rows = [np.full(total, dna.Gap, dtype=np.uint8) for _ in range(len(alpha) + len(beta))]
    acol = bcol = col = 0
    for n, op in route:
        if op in (0, 2):
            for k, f in enumerate(alpha):
                rows[k][col:col + n] = f.Seq[acol:acol + n]
        if op in (0, 1):
            for k, f in enumerate(beta):
                rows[len(alpha) + k][col:col + n] = f.Seq[bcol:bcol + n]
        if op != 1:
            acol += n
        if op != 2:
            bcol += n
        col += n
    return [Fasta(f.Name, rows[k]) for k, f in enumerate(list(alpha) + list(beta))]


def all_seq_affine_oracle(records, scores, gap_open, gap_extend, chunk=1):
    groups = [[r] for r in records]
    while len(groups) > 1:
        best = None
        for x in range(len(groups) - 1):
            for y in range(x + 1, len(groups)):
                A = np.stack([g.Seq for g in groups[x]])
                B = np.stack([g.Seq for g in groups[y]])
                score, route = oracle.multiple_affine_gap(scores, gap_open, gap_extend, chunk, A, B)
                if best is None or score > best[0]:
                    best = (score, x, y, route)
        _, x, y, route = best
        groups[x] = merge(groups[x], groups[y], route)
        groups[y] = groups[-1]
        groups = groups[:-1]
    return groups[0]
