"""cigar.Cigar of the graph aligner: SAM-style ops as bytes (/root/reference/cigar/cigar.go:15-35, cigar/tools.go:58-66).
A different type from align.Cigar (ColType 0/1/2); only what the seed-extension DPs of "next" row N2 need."""

Match, Insertion, Deletion = ord("M"), ord("I"), ord("D")
_FROM_COL = {0: Match, 1: Insertion, 2: Deletion}


class Cigar:
    __slots__ = ("RunLength", "Op")

    def __init__(self, RunLength, Op):
        self.RunLength = int(RunLength)
        self.Op = int(Op)

    def __eq__(self, other):
        return isinstance(other, Cigar) and self.RunLength == other.RunLength and self.Op == other.Op

    def __repr__(self):
        return "{%d %d}" % (self.RunLength, self.Op)  # Go's %v of the struct


def TripleMaxTrace(a, b, c):
    """cigar/tools.go:58-66"""
    if a >= b and a >= c:
        return a, Match
    if b >= c:
        return b, Insertion
    return c, Deletion


def ReverseCigar(alpha):
    """cigar.ReverseCigar: in place."""
    alpha.reverse()


def ToString(cigars):
    """cigar.ToString: 150M style; '*' for an empty slice."""
    if not cigars:
        return "*"
    return "".join("%d%c" % (c.RunLength, c.Op) for c in cigars)


def from_col(op):
    return _FROM_COL[int(op)]
