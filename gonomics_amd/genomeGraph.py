"""Host mirror of the graph aligner's read path ("next" rows N2 and N4 of SURVEY 8f), batched over the GPU:

  N2  LeftDynamicAln / RightDynamicAln               /root/reference/genomeGraph/search.go:234-321   -> gnx_gsw_extend_batch
      LeftAlignTraversal / RightAlignTraversal        search.go:166-232
      GraphSmithWatermanToGiraf                       genomeGraph/toGiraf.go:17-72                    -> GswBatchToGiraf (this file)
  N4  IndexGenomeIntoMap                              genomeGraph/index.go:21-59                      -> gnx_seed_index_build
      seedMapMemPool, extendToTheRightDev / LeftDev   search.go:425-590                               -> gnx_seed_find_batch (+ host for node
      dnaTwoBit.CountRightMatches / CountLeftMatches  dna/dnaTwoBit/perfectAlign.go:10-85                borders), seed_map_host (all host)
      seedCouldBeBetter                               index.go:102-121

The DPs and the seed search (hash lookup + exact-match extension: byte / integer work) run on the device for a whole batch of
reads; what stays here is the reference's bookkeeping: seeds as linked parts across nodes, the per-read loop over sorted seeds with its
`seedCouldBeBetter` pruning, paths, soft clips.  Reads advance in ROUNDS: every read that still has a seed to try contributes its left and
right extension to one gnx_gsw_extend_batch call per side and round (a read's seeds are tried one after the other because each
result moves the pruning bound, exactly like the reference's loop).

PARITY CONTRACT (the reference's own tests of this path only log: UNPINNED):
  * two-bit words are built like dnaTwoBit.BasesToUint64LeftAln does (`answer | uint64(base)`: an N = 4 spills into the low bit
    of the base before it inside a 32-base word), so hits and match lengths are those of the reference also for reads / nodes with N;
  * seed ORDER: <= 100 seeds go through the reference's own heapSortSeeds (search.go:338-370, deterministic, restated literally);
    more than 100 go through Go's sort.Slice, an unstable pdqsort whose order among equal TotalLength this image cannot observe (no
    Go toolchain): here they are sorted by TotalLength descending, ties in order of discovery (stable).  Equal-length seeds that
    reach the same score are interchangeable for the score; which one is reported can differ from Go in that case only;
  * the route a traversal hands from one sibling branch to the next is carried over like in Go (resetDynamicScore is a no-op on
    its by-value argument, search.go:104-107), and so is what Go does THROUGH SHARED BACKING ARRAYS when a node has several Prev /
    Next edges (round 4): a later sibling's DP writing through the route slice an earlier best alignment still points into, in-place
    cigar.Append / ReverseCigar (GoSlice below: header + shared array + Go 1.25's append capacities), and seeds pointing into the
    re-used `nextParts` slice (search.go:447-456: _SeedSlice).  The mirror equals the literal restatement (tests/pyref_gsw.py, which
    models the same slices independently) bit for bit on linear, bubble and three-allele graphs.
"""
import numpy as np

from . import _lib, cigar

# ---------------------------------------------------------------------------------------------------------------------
# N2 primitives (unchanged API)
# ---------------------------------------------------------------------------------------------------------------------


def _merge_route(route_in, runs):
    """The route-building loop of search.go:252-262 / 298-308 applied to the traced ops `runs` = [(len, op)], traceback order."""
    route = [cigar.Cigar(c.RunLength, c.Op) for c in (route_in or [])]
    idx = 0
    for run, op in runs:
        for _ in range(int(run)):
            if len(route) == 0:
                route.append(cigar.Cigar(1, op))
            elif route[idx].Op == op:
                route[idx].RunLength += 1
            else:
                route.append(cigar.Cigar(1, op))
                idx += 1
    return route


# Go slices.  LeftAlignTraversal / RightAlignTraversal hand ONE route slice from sibling to sibling, keep headers of it, reverse it in
# place, and GraphSmithWatermanToGiraf appends to it in place (cigar.Append / Concat): what a later sibling's DP writes through the
# shared backing array shows through every header that still points into it, until an append outgrows the capacity (round 4, VERDICT r3
# missing 1: rounds 1-3 copied lists -- "value semantics" -- and differed from the Go program on ~12 % of the reads of a variant graph).
# The model: (backing array, offset, len, cap) with Go 1.25's growth rule (runtime/slice.go nextslicecap + roundupsize; go.mod: go 1.25).
_GO_SIZE_CLASSES = [0, 8, 16, 24, 32, 48, 64, 80, 96, 112, 128, 144, 160, 176, 192, 208, 224, 240, 256, 288, 320, 352, 384, 416, 448, 480, 512, 576, 640,
                    704, 768, 896, 1024, 1152, 1280, 1408, 1536, 1792, 2048, 2304, 2688, 3072, 3200, 3456, 4096, 4864, 5376, 6144, 6528, 6784, 6912,
                    8192, 9472, 9728, 10240, 10880, 12288, 13568, 14336, 16384, 18432, 19072, 20480, 21760, 24576, 27264, 28672, 32768]


def _go_next_cap(newLen, oldCap, elemSize=16):
    newcap = oldCap
    if newLen > 2 * oldCap:
        newcap = newLen
    elif oldCap < 256:
        newcap = 2 * oldCap
    else:
        while newcap < newLen:
            newcap += (newcap + 3 * 256) >> 2
    mem = newcap * elemSize
    mem = next(c for c in _GO_SIZE_CLASSES if c >= mem) if mem <= 32768 else (mem + 8191) // 8192 * 8192
    return mem // elemSize


class GoSlice:
    """[]cigar.Cigar as Go sees it: a header over a shared backing array of cigar.Cigar cells"""
    __slots__ = ("arr", "off", "n", "cap")

    def __init__(self, arr=None, off=0, n=0, cap=0):
        self.arr, self.off, self.n, self.cap = arr if arr is not None else [], off, n, cap

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        if not 0 <= i < self.n:
            raise IndexError("index out of range [%d] with length %d" % (i, self.n))
        return self.arr[self.off + i]

    def __setitem__(self, i, v):
        if not 0 <= i < self.n:
            raise IndexError("index out of range [%d] with length %d" % (i, self.n))
        self.arr[self.off + i] = v

    def __iter__(self):
        return iter(self.arr[self.off:self.off + self.n])

    def tail(self, k):
        return GoSlice(self.arr, self.off + k, self.n - k, self.cap - k)

    def append(self, *vals):
        """append(s, vals...): in place while the capacity lasts (the cells are copies of the values, as Go copies structs)"""
        need = self.n + len(vals)
        if need <= self.cap:
            for k, v in enumerate(vals):
                self.arr[self.off + self.n + k] = cigar.Cigar(v.RunLength, v.Op)
            return GoSlice(self.arr, self.off, need, self.cap)
        cap = _go_next_cap(need, self.cap)
        arr = [cigar.Cigar(c.RunLength, c.Op) for c in self] + [cigar.Cigar(v.RunLength, v.Op) for v in vals] + [None] * (cap - need)
        return GoSlice(arr, 0, need, cap)

    def reverse(self):  # cigar.ReverseCigar: in place
        i, j = 0, self.n - 1
        while i < self.n // 2:
            a, b = self[i], self[j]
            self[i], self[j] = b, a
            i, j = i + 1, j - 1


def _merge_route_go(route, runs):
    """search.go:252-262 / 298-308 on a Go slice: increments go through the shared array, appends follow Go's capacities"""
    idx = 0
    for run, op in runs:
        for _ in range(int(run)):
            if route.n == 0:
                route = route.append(cigar.Cigar(1, op))
            elif route[idx].Op == op:
                route[idx].RunLength += 1
            else:
                route = route.append(cigar.Cigar(1, op))
                idx += 1
    return route


def DynamicAlnBatch(side, alphas, betas, scores, gapPen, routes=None):
    """side "left" / "right"; returns a list of (score, route, i, j) like the two Go functions."""
    s, ei, ej, ops, off = _lib.gsw_extend_batch(_lib.GNX_GSW_LEFT if side == "left" else _lib.GNX_GSW_RIGHT, scores, gapPen, alphas, betas)
    out = []
    for p in range(len(alphas)):
        runs = [(int(ops["run_length"][k]), cigar.from_col(ops["op"][k])) for k in range(int(off[p]), int(off[p + 1]))]
        rin = routes[p] if routes is not None else None
        if isinstance(rin, GoSlice):  # the traversals' routes: Go slices, merged through their shared arrays
            route = _merge_route_go(rin, runs)
        elif rin:
            route = _merge_route(rin, runs)
        else:
            route = [cigar.Cigar(r, o) for r, o in runs]
        out.append((int(s[p]), route, int(ei[p]), int(ej[p])))
    return out


def LeftDynamicAln(alpha, beta, scores, gapPen, route=None):
    """(score, route, i, j) -- search.go:234-276"""
    return DynamicAlnBatch("left", [alpha], [beta], scores, gapPen, [route])[0]


def RightDynamicAln(alpha, beta, scores, gapPen, route=None):
    """(score, route, maxI, maxJ) -- search.go:278-321"""
    return DynamicAlnBatch("right", [alpha], [beta], scores, gapPen, [route])[0]


# ---------------------------------------------------------------------------------------------------------------------
# graph, two-bit words, reads
# ---------------------------------------------------------------------------------------------------------------------
M64 = (1 << 64) - 1


class TwoBit:
    """dnaTwoBit.TwoBit: 32 bases per uint64, left aligned (dna/dnaTwoBit/dnaTwoBit.go:10-13, 66-76)"""
    __slots__ = ("Seq", "Len")

    def __init__(self, bases):
        b = [int(x) for x in bases]
        self.Len = len(b)
        self.Seq = []
        for start in range(0, len(b), 32):
            w = 0
            for x in b[start:start + 32]:
                w = ((w << 2) | x) & M64  # BasesToUint64LeftAln: an N (4) also sets the low bit of the base before it
            w = (w << (2 * (32 - len(b[start:start + 32])))) & M64
            self.Seq.append(w)


def NewTwoBitRainbow(bases):
    """dna/dnaTwoBit/rainbow.go:27-45: the sequence with 0..31 'A's in front, each as a TwoBit"""
    clone = [int(x) for x in bases]
    out = []
    for _ in range(32):
        out.append(TwoBit(clone))
        clone = [0] + clone
    return out


def _lz64(x):
    return 64 - x.bit_length()


def _tz64(x):
    return 64 if x == 0 else (x & -x).bit_length() - 1


def CountRightMatches(one, startOne, two, startTwo):
    """dna/dnaTwoBit/perfectAlign.go:10-49"""
    offsetOne, offsetTwo = (startOne % 32) * 2, (startTwo % 32) * 2
    if offsetOne != offsetTwo:
        raise ValueError("Error: Different offsets when comparing sequences")
    i, j = startOne // 32, startTwo // 32
    iEnd, jEnd = (one.Len + 31) // 32, (two.Len + 31) // 32
    seqDiff = (one.Seq[i] ^ two.Seq[j]) & (M64 >> offsetOne)
    bitMatches = _lz64(seqDiff)
    total = bitMatches - offsetOne
    i, j = i + 1, j + 1
    while i < iEnd and j < jEnd and bitMatches == 64:
        bitMatches = _lz64(one.Seq[i] ^ two.Seq[j])
        total += bitMatches
        i, j = i + 1, j + 1
    return min(total // 2, one.Len - startOne, two.Len - startTwo)


def CountLeftMatches(one, startOne, two, startTwo):
    """dna/dnaTwoBit/perfectAlign.go:51-85"""
    offsetOne, offsetTwo = (startOne % 32) * 2, (startTwo % 32) * 2
    if offsetOne != offsetTwo:
        raise ValueError("Different offsets when comparing sequences")
    firstBitsNoLook = 64 - offsetOne - 2
    i, j = startOne // 32, startTwo // 32
    seqDiff = (one.Seq[i] ^ two.Seq[j]) & ((M64 << firstBitsNoLook) & M64)
    bitMatches = _tz64(seqDiff)
    total = bitMatches - firstBitsNoLook
    i, j = i - 1, j - 1
    while i >= 0 and j >= 0 and bitMatches == 64:
        bitMatches = _tz64(one.Seq[i] ^ two.Seq[j])
        total += bitMatches
        i, j = i - 1, j - 1
    return total // 2


def GetBase(frag, pos):
    """dnaTwoBit.GetBase"""
    return (frag.Seq[pos // 32] >> (64 - 2 * (pos % 32 + 1))) & 3


class Edge:
    __slots__ = ("Dest", "Prob")

    def __init__(self, Dest, Prob=1.0):
        self.Dest, self.Prob = Dest, Prob


class Node:
    """genomeGraph.Node (genomeGraph.go:25-33)"""
    __slots__ = ("Id", "Seq", "SeqTwoBit", "Prev", "Next")

    def __init__(self, Id, Seq):
        self.Id = int(Id)
        self.Seq = np.ascontiguousarray(Seq, dtype=np.uint8)
        if self.Seq.size and int(self.Seq.max()) > 4:
            raise ValueError("node bases must be A, C, G, T or N (0..4)")
        self.SeqTwoBit = TwoBit(self.Seq)
        self.Prev, self.Next = [], []


class GenomeGraph:
    def __init__(self):
        self.Nodes = []


def AddNode(g, n):
    if n.Id != len(g.Nodes):
        raise ValueError("nodes are added in Id order")
    g.Nodes.append(n)
    return n


def AddEdge(u, v, p=1.0):
    """genomeGraph.go:118-121"""
    u.Next.append(Edge(v, p))
    v.Prev.append(Edge(u, p))


_COMP = np.asarray([3, 2, 1, 0, 4], dtype=np.uint8)


class FastqBig:
    """fastq.FastqBig (fastq/fastqBig.go:15-50)"""

    def __init__(self, Name, Seq, Qual=None):
        self.Name = Name
        self.Seq = np.ascontiguousarray(Seq, dtype=np.uint8)
        if self.Seq.size and int(self.Seq.max()) > 4:
            raise ValueError("read bases must be A, C, G, T or N (0..4)")
        self.SeqRc = np.ascontiguousarray(_COMP[self.Seq][::-1])
        self.Qual = Qual
        self.Rainbow = None
        self.RainbowRc = None

    def rainbows(self):
        if self.Rainbow is None:
            self.Rainbow, self.RainbowRc = NewTwoBitRainbow(self.Seq), NewTwoBitRainbow(self.SeqRc)
        return self.Rainbow, self.RainbowRc


def ChromAndPosToNumber(chrom, start):
    return (int(chrom) << 32) | int(start)


def numberToChromAndPos(code):
    return int(code) >> 32, int(code) & 0xFFFFFFFF


def dnaToNumber(seq, start, end):
    """align.go:170-177 (`answer << 2 | base`: an N spills like in the two-bit words; IndexGenomeIntoMap skips N k-mers anyway)"""
    ans = int(seq[start])
    for i in range(start + 1, end):
        ans = ((ans << 2) | int(seq[i])) & M64
    return ans


# ---------------------------------------------------------------------------------------------------------------------
# N4: index (host statement; the device builds the same (key, location) lists with gnx_seed_index_build)
# ---------------------------------------------------------------------------------------------------------------------
def IndexGenomeIntoMap(genome, seedLen, seedStep):
    """index.go:21-59: {k-mer code: [location codes in insertion order]}"""
    if seedLen < 2 or seedLen > 32:
        raise ValueError("Error: seed length needs to be greater than 1 and less than 33.  Got: %d" % seedLen)
    answer = {}

    def helper(prevSeq, currNode, locationCode):
        if len(prevSeq) + len(currNode.Seq) >= seedLen:
            currSeq = list(prevSeq) + [int(x) for x in currNode.Seq[0:seedLen - len(prevSeq)]]
            if 4 not in currSeq:
                answer.setdefault(dnaToNumber(currSeq, 0, seedLen), []).append(locationCode)
        else:
            for e in currNode.Next:
                helper(list(prevSeq) + [int(x) for x in currNode.Seq], e.Dest, locationCode)

    for nodeIdx, n in enumerate(genome):
        seq = n.Seq
        pos = 0
        while pos < len(seq) - seedLen + 1:
            if not (seq[pos:pos + seedLen] == 4).any():
                answer.setdefault(dnaToNumber(seq, pos, pos + seedLen), []).append(ChromAndPosToNumber(nodeIdx, pos))
            pos += seedStep
        while pos < len(seq):
            loc = ChromAndPosToNumber(nodeIdx, pos)
            for e in n.Next:
                helper([int(x) for x in seq[pos:]], e.Dest, loc)
            pos += seedStep
    return answer


class SeedIndex:
    """The same index as two sorted arrays (keys ascending, stable: locations of one key in insertion order), built on the device
    for the k-mers inside nodes (gnx_seed_index_build: one thread per position, radix sort) and merged with the few k-mers that
    cross node borders (host recursion over Next edges, like indexGenomeIntoMapHelper)."""

    def __init__(self, genome, seedLen, seedStep, device=True):
        self.seedLen, self.seedStep = seedLen, seedStep
        if device:
            keys, locs = _lib.seed_index_build([n.Seq for n in genome], seedLen, seedStep)
        else:
            keys, locs = [], []
        order = []  # (key, insertion rank) for the border k-mers + (host mode) all k-mers
        if not device:
            full = IndexGenomeIntoMap(genome, seedLen, seedStep)
            ks = sorted(full)
            self.keys = np.asarray([k for k in ks for _ in full[k]], dtype=np.uint64)
            self.locs = np.asarray([v for k in ks for v in full[k]], dtype=np.uint64)
            return
        border = {}

        def helper(prevSeq, currNode, locationCode):
            if len(prevSeq) + len(currNode.Seq) >= seedLen:
                currSeq = list(prevSeq) + [int(x) for x in currNode.Seq[0:seedLen - len(prevSeq)]]
                if 4 not in currSeq:
                    border.setdefault(dnaToNumber(currSeq, 0, seedLen), []).append(locationCode)
            else:
                for e in currNode.Next:
                    helper(list(prevSeq) + [int(x) for x in currNode.Seq], e.Dest, locationCode)

        for nodeIdx, n in enumerate(genome):
            L = len(n.Seq)
            first = 0 if L - seedLen + 1 <= 0 else ((L - seedLen) // seedStep + 1) * seedStep
            for pos in range(first, L, seedStep):
                for e in n.Next:
                    helper([int(x) for x in n.Seq[pos:]], e.Dest, ChromAndPosToNumber(nodeIdx, pos))
        if border:
            # insertion order of the reference: by node, inside nodes first, then the node's border positions -> a location code
            # (node << 32 | pos) sorts exactly like that within one key, because border positions of a node come after its inner ones
            bk = np.asarray([k for k in border for _ in border[k]], dtype=np.uint64)
            bl = np.asarray([v for k in border for v in border[k]], dtype=np.uint64)
            keys = np.concatenate([keys, bk])
            locs = np.concatenate([locs, bl])
            o = np.lexsort((locs, keys))
            keys, locs = keys[o], locs[o]
        self.keys, self.locs = keys, locs

    def lookup(self, key):
        lo = int(np.searchsorted(self.keys, np.uint64(key), side="left"))
        hi = int(np.searchsorted(self.keys, np.uint64(key), side="right"))
        return [int(x) for x in self.locs[lo:hi]]


# ---------------------------------------------------------------------------------------------------------------------
# N4: seeds
# ---------------------------------------------------------------------------------------------------------------------
class SeedDev:
    """index.go:10-18"""
    __slots__ = ("TargetId", "TargetStart", "QueryStart", "Length", "PosStrand", "TotalLength", "NextPart")

    def __init__(self, TargetId, TargetStart, QueryStart, Length, PosStrand, TotalLength, NextPart=None):
        self.TargetId, self.TargetStart, self.QueryStart, self.Length = int(TargetId), int(TargetStart), int(QueryStart), int(Length)
        self.PosStrand, self.TotalLength, self.NextPart = bool(PosStrand), int(TotalLength), NextPart

    def key(self):
        """the seed as a tuple of its parts (for comparisons in tests)"""
        out, s = [], self
        while s is not None:
            out.append((s.TargetId, s.TargetStart, s.QueryStart, s.Length, s.PosStrand, s.TotalLength))
            s = s.NextPart
        return tuple(out)


def getLastPart(a):
    while a.NextPart is not None:
        a = a.NextPart
    return a


def getSeedPath(seed):
    path = [seed.TargetId]
    if seed.NextPart is not None:
        path += getSeedPath(seed.NextPart)
    return path


class _SeedSlice:
    """[]SeedDev as extendToTheRightDev uses it: the cells are SeedDev objects with identity (NextPart pointers point AT cells), storing a
    value into a cell overwrites the cell's fields; append follows Go's capacities (SeedDev: 32 bytes)"""
    __slots__ = ("arr", "n", "cap")

    def __init__(self, arr=None, n=0, cap=0):
        self.arr, self.n, self.cap = arr if arr is not None else [], n, cap

    def append(self, v):
        if self.n < self.cap:
            c = self.arr[self.n]  # in place: whoever points at this cell sees the new seed
            c.TargetId, c.TargetStart, c.QueryStart, c.Length, c.PosStrand, c.TotalLength, c.NextPart = v.TargetId, v.TargetStart, v.QueryStart, v.Length, v.PosStrand, v.TotalLength, v.NextPart
            return _SeedSlice(self.arr, self.n + 1, self.cap)
        cap = _go_next_cap(self.n + 1, self.cap, 32)
        cp = lambda x: SeedDev(x.TargetId, x.TargetStart, x.QueryStart, x.Length, x.PosStrand, x.TotalLength, x.NextPart)  # noqa: E731
        arr = [cp(self.arr[k]) for k in range(self.n)] + [cp(v)] + [SeedDev(0, 0, 0, 0, True, 0, None) for _ in range(cap - self.n - 1)]
        return _SeedSlice(arr, self.n + 1, cap)

    def values(self):
        """copies of the values (what `append(finalSeeds, tempSeeds...)` and `for _, tempSeed = range tempSeeds` take)"""
        return [SeedDev(x.TargetId, x.TargetStart, x.QueryStart, x.Length, x.PosStrand, x.TotalLength, x.NextPart) for x in self.arr[:self.n]]


def _extend_right(node, read, readStart, nodeStart, posStrand, answer):
    """extendToTheRightDev (search.go:425-461) with its slice re-use: `nextParts` is handed back into the call for the next edge of
    node.Next and overwritten in place there (`answer = answer[:0]`), while the seeds made for the edge before still point at its cells
    (`NextPart: &nextParts[j]`).  Returns a _SeedSlice or None (Go: nil)."""
    answer = _SeedSlice(answer.arr, 0, answer.cap) if answer is not None else _SeedSlice()
    rb, rbrc = read.rainbows()
    nodeOffset = nodeStart % 32
    readOffset = 31 - ((readStart - nodeOffset + 31) % 32)
    rightMatches = CountRightMatches(node.SeqTwoBit, nodeStart, (rb if posStrand else rbrc)[readOffset], readStart + readOffset)
    if rightMatches == 0:
        return None
    nextParts = None
    if readStart + rightMatches < len(read.Seq) and nodeStart + rightMatches == node.SeqTwoBit.Len and len(node.Next) != 0:
        for e in node.Next:
            nextParts = _extend_right(e.Dest, read, readStart + rightMatches, 0, posStrand, nextParts)
            for j in range(nextParts.n if nextParts is not None else 0):
                cell = nextParts.arr[j]
                answer = answer.append(SeedDev(node.Id, nodeStart, readStart, rightMatches, posStrand, rightMatches + cell.TotalLength, cell))
    if answer.n == 0:
        answer = _SeedSlice([SeedDev(node.Id, nodeStart, readStart, rightMatches, posStrand, rightMatches, None)], 1, 1)
    return answer


def extendToTheRightDev(node, read, readStart, nodeStart, posStrand):
    """search.go:425-461: the seeds (values) that start at (node, nodeStart) / readStart and run to the right, across node borders"""
    out = _extend_right(node, read, readStart, nodeStart, posStrand, None)
    return out.values() if out is not None else []


def _left_helper(node, read, nextPart):
    """search.go:488-531"""
    rb, rbrc = read.rainbows()
    nodePos = node.SeqTwoBit.Len - 1
    readPos = nextPart.QueryStart - 1
    nodeOffset = nodePos % 32
    readOffset = 31 - ((readPos - nodeOffset + 31) % 32)
    leftMatches = min(readPos + 1, CountLeftMatches(node.SeqTwoBit, nodePos, (rb if nextPart.PosStrand else rbrc)[readOffset], readPos + readOffset))
    if leftMatches == 0:
        raise RuntimeError("Error: should not have zero matches to the left")
    currPart = SeedDev(node.Id, nodePos - (leftMatches - 1), readPos - (leftMatches - 1), leftMatches, nextPart.PosStrand, leftMatches + nextPart.TotalLength,
                       nextPart)  # (`NextPart: &nextPart`: the callee's own copy of its argument -- every caller passes a value copy)
    answer = []
    if currPart.QueryStart > 0 and currPart.TargetStart == 0:
        for e in node.Prev:
            readBase = GetBase((rb if nextPart.PosStrand else rbrc)[0], currPart.QueryStart - 1)
            if readBase == GetBase(e.Dest.SeqTwoBit, e.Dest.SeqTwoBit.Len - 1):
                answer += _left_helper(e.Dest, read, currPart)
    return answer if answer else [currPart]


def extendToTheLeftDev(node, read, currPart):
    """search.go:463-486"""
    rb, rbrc = read.rainbows()
    answer = []
    if currPart.QueryStart > 0 and currPart.TargetStart == 0:
        for e in node.Prev:
            readBase = GetBase((rb if currPart.PosStrand else rbrc)[0], currPart.QueryStart - 1)
            if readBase == GetBase(e.Dest.SeqTwoBit, e.Dest.SeqTwoBit.Len - 1):
                answer += _left_helper(e.Dest, read, currPart)
    return answer if answer else [currPart]


def heapSortSeeds(a):
    """search.go:338-370, literally (a min-heap on TotalLength popped to the back: descending order, its own tie order)"""
    def heapify(a, n, i):
        while True:
            l, r = 2 * i + 1, 2 * i + 2
            mx = l if (l < n and a[l].TotalLength < a[i].TotalLength) else i
            if r < n and a[r].TotalLength < a[mx].TotalLength:
                mx = r
            if mx == i:
                return
            a[i], a[mx] = a[mx], a[i]
            i = mx

    n = len(a)
    for i in range(n // 2 - 1, -1, -1):
        heapify(a, n, i)
    size = n
    for i in range(n - 1, 0, -1):
        a[0], a[i] = a[i], a[0]
        size -= 1
        heapify(a, size, 0)


# census of the situations in which the parity contract has a declared deviation (tools/bench_gsw.py reports it): reads whose seed list
# goes through Go's unstable sort.Slice, traversals that branch (where Go's shared backing arrays can alias), reads Go panics on
STATS = {"reads_with_more_than_100_seeds": 0, "seed_lists": 0, "branching_left_traversals": 0, "branching_right_traversals": 0}


def sort_seeds(seeds):
    """seedMapMemPool's tail (search.go:583-589); see the parity contract for > 100 seeds"""
    STATS["seed_lists"] += 1
    if len(seeds) > 100:
        STATS["reads_with_more_than_100_seeds"] += 1
        seeds.sort(key=lambda s: -s.TotalLength)  # Python's sort is stable: ties stay in order of discovery
    else:
        heapSortSeeds(seeds)
    return seeds


def _read_key(rainbow, readStart, seedLen):
    keyIdx = (readStart + 31) // 32
    keyOffset = 31 - ((readStart + 31) % 32)
    return rainbow[keyOffset].Seq[keyIdx] >> (64 - 2 * seedLen)


def seed_map_host(index, nodes, read, seedLen):
    """seedMapMemPool (search.go:549-590) on the host: `index` is a dict (IndexGenomeIntoMap) or a SeedIndex"""
    rb, rbrc = read.rainbows()
    look = index.lookup if isinstance(index, SeedIndex) else (lambda k: index.get(k, []))
    final = []
    for readStart in range(0, len(read.Seq) - seedLen + 1):
        for pos_strand, rain in ((True, rb), (False, rbrc)):
            for code in look(_read_key(rain, readStart, seedLen)):
                nodeIdx, nodePos = numberToChromAndPos(code)
                nodeOffset = nodePos % 32
                readOffset = 31 - ((readStart - nodeOffset + 31) % 32)
                left = min(readStart + 1, CountLeftMatches(nodes[nodeIdx].SeqTwoBit, nodePos, rain[readOffset], readStart + readOffset))
                temp = extendToTheRightDev(nodes[nodeIdx], read, readStart - (left - 1), nodePos - (left - 1), pos_strand)
                if pos_strand:
                    for t in temp:
                        final += extendToTheLeftDev(nodes[nodeIdx], read, t)
                else:
                    final += temp  # (the reference does not extend minus-strand seeds to the left across nodes)
    return sort_seeds(final)


def seed_map_batch(index, nodes, reads, seedLen):
    """seedMapMemPool for a batch of reads with the hash lookups and the in-node exact-match extensions on the device
    (gnx_seed_find_batch: one thread per (read, strand, position)); parts that continue into neighbouring nodes are finished here."""
    hits = _lib.seed_find_batch(index.keys, index.locs, [n.Seq for n in nodes], [r.Seq for r in reads], seedLen)
    out = []
    for r, read in enumerate(reads):
        final = []
        for (readStart, strand, nodeIdx, nodeStart, qStart, right) in hits[r]:  # in the reference's order of discovery
            node = nodes[nodeIdx]
            pos_strand = strand == 0
            if right == 0:
                continue
            if qStart + right < len(read.Seq) and nodeStart + right == node.SeqTwoBit.Len and len(node.Next) != 0:
                temp = extendToTheRightDev(node, read, qStart, nodeStart, pos_strand)  # crosses into the next node(s)
            else:
                temp = [SeedDev(node.Id, nodeStart, qStart, right, pos_strand, right, None)]
            if pos_strand:
                for t in temp:
                    final += extendToTheLeftDev(node, read, t)
            else:
                final += temp
        out.append(sort_seeds(final))
    return out


def seedCouldBeBetter(seedLen, currBestScore, perfectScore, queryLen, maxMatch, minMatch, leastSevereMismatch, leastSevereMatchMismatchChange):
    """index.go:102-121 (Go integer division and remainder: operands are non-negative here)"""
    seeds = queryLen // (seedLen + 1)
    remainder = queryLen % (seedLen + 1)
    if seedLen * maxMatch >= currBestScore and perfectScore - ((queryLen - seedLen) * minMatch) >= currBestScore:
        return True
    if seedLen * seeds * maxMatch + seeds * leastSevereMismatch >= currBestScore and \
            perfectScore - remainder * minMatch + seeds * leastSevereMatchMismatchChange >= currBestScore:
        return True
    if seedLen * seeds * maxMatch + remainder * maxMatch + (seeds + 1) * leastSevereMismatch >= currBestScore and \
            perfectScore + (seeds + 1) * leastSevereMatchMismatchChange >= currBestScore:
        return True
    return False


# ---------------------------------------------------------------------------------------------------------------------
# N2: traversals and the per-read driver, as coroutines that yield their DP requests
# ---------------------------------------------------------------------------------------------------------------------
def AddPath(allPaths, newPath):
    if len(allPaths) == 0 or allPaths[-1] != newPath:
        allPaths.append(newPath)
    return allPaths


def CatPaths(currPaths, newPaths):
    if len(newPaths) == 0:
        return currPaths
    if len(currPaths) == 0:
        return newPaths
    currPaths = AddPath(currPaths, newPaths[0])
    return currPaths + list(newPaths[1:])


class GoPanic(RuntimeError):
    """a situation in which the Go code panics (the process dies): reported, never papered over"""


def _left_target(n, extension, refEnd, seq):
    """getLeftTargetBases (search.go:135-140), the expression AS WRITTEN: `refEnd-numbers.Min(len(seq)+refEnd, extension)-len(seq)` is
    left-associative, i.e. the slice starts at refEnd - min(..) - len(seq) -- not at refEnd - (min(..) - len(seq)) as the commented-out
    lines above it intend.  With an empty `seq` (the first node of a traversal) both agree.  With bases already collected (a
    traversal that went into a Prev node) the reference takes extension + len(seq) bases of that node when it is long enough, and
    panics with a negative slice bound when it is not."""
    start = refEnd - min(len(seq) + refEnd, extension) - len(seq)
    if start < 0 or start > refEnd:
        raise GoPanic("runtime error: slice bounds out of range [%d:%d] (getLeftTargetBases, search.go:139: node %d, %d bases collected, extension %d)"
                      % (start, refEnd, n.Id, len(seq), extension))
    return np.concatenate([n.Seq[start:refEnd], seq]).astype(np.uint8)


def _right_target(n, extension, start, seq):
    """getRightBases (search.go:142-147)"""
    take = min(len(seq) + len(n.Seq) - start, extension) - len(seq)
    return np.concatenate([seq, n.Seq[start:start + take]]).astype(np.uint8)


def _left_traversal(n, seq, refEnd, currentPath, extension, read, route):
    """LeftAlignTraversal (search.go:166-198) as a generator: yields ("left", target, read, route_in), receives (score, route, i, j);
    returns (alignment, score, targetStart, queryStart, path)"""
    sSeq = _left_target(n, extension, refEnd, seq)
    sPath = list(currentPath)  # search.go:174-176 calls AddPath(s.Path, n.Id) and drops its result; s.Path was made with len == cap,
    #                            so the append went to a new array: the node is never recorded (left and right paths stay empty)
    if len(seq) + refEnd >= extension or len(n.Prev) == 0:
        score, aln, tStart, qStart = yield ("left", sSeq, read, route)
        return aln, score, refEnd - len(sSeq) - len(seq) + tStart, qStart, sPath
    best = None
    leftScore = -(1 << 63)
    if len(n.Prev) > 1:
        STATS["branching_left_traversals"] += 1
    for e in n.Prev:
        route, cs, ts, qs, cpath = yield from _left_traversal(e.Dest, sSeq, len(e.Dest.Seq), sPath, extension, read, route)
        if cs > leftScore:
            leftScore = cs
            best = (route, refEnd - len(sSeq) - len(seq) + ts, qs, cpath)  # the slice HEADER: later siblings write through the same array
    aln, tStart, qStart, path = best
    aln.reverse()  # cigar.ReverseCigar(sk.leftAlignment): in place
    path = list(reversed(path))
    return aln, leftScore, tStart, qStart, path


def _right_traversal(n, seq, start, currentPath, extension, read, route):
    """RightAlignTraversal (search.go:200-232)"""
    sSeq = _right_target(n, extension, start, seq)
    sPath = list(currentPath)
    if len(seq) + len(n.Seq) - start >= extension or len(n.Next) == 0:
        score, aln, tEnd, qEnd = yield ("right", sSeq, read, route)
        return aln, score, tEnd + start, qEnd, sPath
    best = None
    rightScore = -(1 << 63)
    if len(n.Next) > 1:
        STATS["branching_right_traversals"] += 1
    for e in n.Next:
        route, cs, te, qe, cpath = yield from _right_traversal(e.Dest, sSeq, 0, sPath, extension, read, route)
        if cs > rightScore:
            rightScore = cs
            best = (route, te, qe, cpath)
    aln, tEnd, qEnd, path = best
    aln.reverse()
    return aln, rightScore, tEnd + start, qEnd, path


def _query_length(cigs):
    return sum(c.RunLength for c in cigs if c.Op in (cigar.Match, cigar.Insertion, ord("S"), ord("="), ord("X")))


def _append_soft_clips(front, lengthOfRead, cigs):
    """cigar.AppendSoftClips (cigar/tools.go:26-40), literally -- including that a front clip without a back clip returns only the clip"""
    run = _query_length(cigs)
    if front == 0 and run >= lengthOfRead:
        return cigs
    answer = GoSlice([None] * (len(cigs) + 2), 0, 0, len(cigs) + 2)  # make([]Cigar, 0, len(cigars)+2)
    if front > 0:
        answer = answer.append(cigar.Cigar(front, ord("S")))
    if front + run < lengthOfRead:
        answer = answer.append(*list(cigs)).append(cigar.Cigar(lengthOfRead - front - run, ord("S")))
    return answer


def _cig_append(alpha, beta):
    """cigar.Append (cigar/tools.go:4-11) on a Go slice: the last cell is incremented IN the shared array, or beta is appended"""
    if len(alpha) > 0 and alpha[len(alpha) - 1].Op == beta.Op:
        alpha[len(alpha) - 1].RunLength += beta.RunLength
    else:
        alpha = alpha.append(beta)
    return alpha


def _cig_concat(alpha, beta):
    """cigar.Concat (cigar/tools.go:14-23)"""
    if len(alpha) == 0:
        return beta
    if len(beta) > 0:
        alpha = _cig_append(alpha, beta[0])
        beta = beta.tail(1)
    return alpha.append(*list(beta))


class Giraf:
    """giraf.Giraf, the fields GraphSmithWatermanToGiraf fills (giraf/giraf.go:16-33)"""

    def __init__(self, read):
        self.QName, self.QStart, self.QEnd, self.PosStrand = read.Name, 0, 0, True
        self.Path = (0, [], 0)  # (TStart, Nodes, TEnd)
        self.Cigar, self.AlnScore, self.MapQ, self.Seq = None, 0, 255, read.Seq
        self.Flag = 0  # set by WrapPairGirafBatch only (toGiraf.go:130-140)

    def key(self):
        return (self.QStart, self.QEnd, self.PosStrand, self.Path[0], tuple(self.Path[1]), self.Path[2],
                None if self.Cigar is None else tuple((c.RunLength, c.Op) for c in self.Cigar), self.AlnScore, bytes(self.Seq))


def _read_to_giraf(gg, read, seeds, scoreMatrix):
    """GraphSmithWatermanToGiraf (toGiraf.go:17-72) for one read, as a generator of DP requests"""
    sc = np.asarray(scoreMatrix, dtype=np.int64)
    best = Giraf(read)
    perfect = int(sc[read.Seq, read.Seq].sum())
    extension = perfect // 600 + len(read.Seq)
    # scoreKeeper fields that survive from one seed to the next (resetScoreKeeper gets its argument by value: a no-op): a seed that
    # covers the whole read re-uses the alignments, paths and queryEnd of the seed before it (toGiraf.go:47-51)
    leftAln, rightAln, leftPath, rightPath, queryEnd = GoSlice(), GoSlice(), [], [], 0
    for seed in seeds:
        if not seedCouldBeBetter(seed.TotalLength, best.AlnScore, perfect, len(read.Seq), 100, 90, -196, -296):
            break
        tail = getLastPart(seed)
        currSeq = read.Seq if seed.PosStrand else read.SeqRc
        seedScore = int(sc[currSeq[seed.QueryStart:tail.QueryStart + tail.Length], currSeq[seed.QueryStart:tail.QueryStart + tail.Length]].sum())
        if seed.TotalLength == len(currSeq):
            targetStart, targetEnd, queryStart, currScore = seed.TargetStart, tail.TargetStart + tail.Length, seed.QueryStart, seedScore
        else:
            ext = extension - seed.TotalLength
            leftAln, leftScore, targetStart, queryStart, leftPath = yield from _left_traversal(
                gg.Nodes[seed.TargetId], np.zeros(0, np.uint8), seed.TargetStart, [], ext, currSeq[:seed.QueryStart], GoSlice())
            rightAln, rightScore, targetEnd, queryEnd, rightPath = yield from _right_traversal(
                gg.Nodes[tail.TargetId], np.zeros(0, np.uint8), tail.TargetStart + tail.Length, [], ext, currSeq[tail.QueryStart + tail.Length:], GoSlice())
            currScore = leftScore + seedScore + rightScore
        if currScore > best.AlnScore:
            best.QStart = queryStart
            best.QEnd = seed.QueryStart + queryStart + queryEnd + seed.TotalLength - 1
            best.PosStrand = seed.PosStrand
            best.Path = (targetStart, CatPaths(CatPaths(list(leftPath), getSeedPath(seed)), list(rightPath)), targetEnd)
            # (on sk.leftAlignment's own array, like the Go code: a later seed that covers the whole read re-uses the stale header)
            best.Cigar = _append_soft_clips(queryStart, len(currSeq), _cig_concat(_cig_append(leftAln, cigar.Cigar(seed.TotalLength, cigar.Match)), rightAln))
            best.AlnScore = currScore
            best.Seq = currSeq
    if best.Cigar is not None:
        best.Cigar = [cigar.Cigar(c.RunLength, c.Op) for c in best.Cigar]  # what the caller sees when the function returns
    return best


def GswBatchToGiraf(gg, reads, index, seedLen, scoreMatrix, device_seeds=True, on_panic="raise"):
    """GraphSmithWatermanToGiraf for a batch of reads: seeds from the device (or the host statement), then rounds of batched DPs.
    on_panic: what to do with a read on which the Go code panics (getLeftTargetBases with a short Prev node, see _left_target):
    "raise" (default: the Go process would die there) or "mark" (that read's result is the GoPanic instance, the others go on)."""
    seeds = seed_map_batch(index, gg.Nodes, reads, seedLen) if device_seeds else [seed_map_host(index, gg.Nodes, r, seedLen) for r in reads]
    gens = [_read_to_giraf(gg, r, s, scoreMatrix) for r, s in zip(reads, seeds)]
    results = [None] * len(reads)
    pending = {}

    def advance(k, value, first):
        try:
            pending[k] = next(gens[k]) if first else gens[k].send(value)
        except StopIteration as st:
            results[k] = st.value
            pending.pop(k, None)
        except GoPanic as gp:
            if on_panic != "mark":
                raise
            results[k] = gp
            pending.pop(k, None)

    for k in range(len(gens)):
        advance(k, None, True)
    while pending:
        for side in ("left", "right"):
            ks = [k for k, rq in pending.items() if rq[0] == side]
            if not ks:
                continue
            outs = DynamicAlnBatch(side, [pending[k][1] for k in ks], [pending[k][2] for k in ks], scoreMatrix, -600, [pending[k][3] for k in ks])
            for k, o in zip(ks, outs):
                advance(k, o, False)
    return results


def _edge_list(gg):
    """(u, v) pairs in AN order of AddEdge calls that rebuilds every node's Next AND Prev list as they are (traversals try a node's edges
    in list order, so the order is part of the graph): a topological order of the edges under "before its successor in u's Next list" and
    "before its successor in v's Prev list".  Lists that no sequence of AddEdge calls produces are refused."""
    ids = {id(n): k for k, n in enumerate(gg.Nodes)}
    edges = []            # (u, v, occurrence) in Next-list order of the nodes
    slot = {}
    for u in gg.Nodes:
        seen = {}
        for e in u.Next:
            key = (ids[id(u)], ids[id(e.Dest)])
            seen[key] = seen.get(key, 0) + 1
            slot[key + (seen[key],)] = len(edges)
            edges.append(key + (seen[key],))
    succ = [[] for _ in edges]
    indeg = [0] * len(edges)
    for x in range(1, len(edges)):  # Next-list order inside one node
        if edges[x][0] == edges[x - 1][0]:
            succ[x - 1].append(x); indeg[x] += 1
    n_prev = 0
    for v in gg.Nodes:
        seen, last = {}, None
        for e in v.Prev:
            key = (ids[id(e.Dest)], ids[id(v)])
            seen[key] = seen.get(key, 0) + 1
            x = slot.get(key + (seen[key],))
            if x is None:
                raise ValueError("a Prev edge %d -> %d has no Next edge" % key)
            if last is not None:
                succ[last].append(x); indeg[x] += 1
            last = x
            n_prev += 1
    if n_prev != len(edges):
        raise ValueError("the graph's Next and Prev lists hold different edges")
    import heapq
    ready = [x for x in range(len(edges)) if indeg[x] == 0]
    heapq.heapify(ready)
    out = []
    while ready:
        x = heapq.heappop(ready)  # (the smallest ready edge: graphs built node by node come back in their original order)
        out.append(edges[x][:2])
        for y in succ[x]:
            indeg[y] -= 1
            if indeg[y] == 0:
                heapq.heappush(ready, y)
    if len(out) != len(edges):
        raise ValueError("the graph's Next and Prev lists do not come from one sequence of AddEdge calls: pass edges= explicitly")
    return out


class NativeGraph:
    """the graph behind gnx_gsw_graph_create (nodes, edges, seed index kept by the library, the index resident on the device): the whole
    read path of a batch in one C-ABI call -- the per-read driver is the library's compiled one (include/gonomics_genomegraph.hpp on a
    pool of host threads) instead of this module's generators.  Same results as GswBatchToGiraf / WrapPairGirafBatch."""

    def __init__(self, gg, seedLen, seedStep, edges=None):
        self.gg = gg
        self.handle = _lib.GswGraph([n.Seq for n in gg.Nodes], _edge_list(gg) if edges is None else edges, seedLen, seedStep)

    def _girafs(self, reads, scoreMatrix, paired, threads, on_panic):
        gir, nodes, cig = self.handle.map_reads([r.Seq for r in reads], scoreMatrix, -600, paired=paired, threads=threads)
        out = []
        for k, r in enumerate(reads):
            x = gir[k]
            if x["panicked"]:
                gp = GoPanic("runtime error: slice bounds out of range (getLeftTargetBases, search.go:139)")
                if on_panic != "mark":
                    raise gp
                out.append(gp)
                continue
            g = Giraf(r)
            g.QStart, g.QEnd, g.PosStrand, g.AlnScore, g.MapQ, g.Flag = int(x["q_start"]), int(x["q_end"]), bool(x["pos_strand"]), int(x["aln_score"]), int(x["map_q"]), int(x["flag"])
            g.Path = (int(x["t_start"]), [int(v) for v in nodes[int(x["node_off"]):int(x["node_off"] + x["n_nodes"])]], int(x["t_end"]))
            if x["has_cigar"]:
                c = cig[int(x["cigar_off"]):int(x["cigar_off"] + x["n_cigar"])]
                g.Cigar = [cigar.Cigar(int(a), int(b)) for a, b in zip(c["run_length"], c["op"])]
            g.Seq = r.SeqRc if x["seq_is_rc"] else r.Seq
            out.append(g)
        return out

    def GswBatchToGiraf(self, reads, scoreMatrix, threads=0, on_panic="raise"):
        return self._girafs(reads, scoreMatrix, False, threads, on_panic)

    def WrapPairGirafBatch(self, pairs, scoreMatrix, threads=0, on_panic="raise"):
        flat = self._girafs([r for pr in pairs for r in pr], scoreMatrix, True, threads, on_panic)
        return [(flat[2 * k], flat[2 * k + 1]) for k in range(len(pairs))]

    def map_reads_raw(self, read_seqs, scoreMatrix, paired=False, threads=0):
        """the arrays as the C ABI returns them (no per-read Python objects): (GIRAF_DTYPE records, node ids, cigars)"""
        return self.handle.map_reads(read_seqs, scoreMatrix, -600, paired=paired, threads=threads)


def getGirafFlags(ag):
    """toGiraf.go:183-192 (uint8)"""
    return (4 if ag.PosStrand else 0) + (2 if ag.AlnScore < 1200 else 0)


def isProperPairAlign(fwd, rev):
    """toGiraf.go:171-181"""
    if abs(float(fwd.Path[0] - rev.Path[0])) < 10000:
        if fwd.Path[0] < rev.Path[0] and fwd.PosStrand and not rev.PosStrand:
            return True
        if fwd.Path[0] > rev.Path[0] and not fwd.PosStrand and rev.PosStrand:
            return True
    return False


def WrapPairGirafBatch(gg, pairs, index, seedLen, scoreMatrix, device_seeds=True, on_panic="raise"):
    """WrapPairGiraf (toGiraf.go:117-128) for a batch of read pairs [(fwd FastqBig, rev FastqBig)]: both mates of every pair go through
    ONE GswBatchToGiraf call (2 x len(pairs) reads: the same seed search and DP rounds), then setGirafFlags (toGiraf.go:130-140) as
    written -- the forward mate gets +8 and +16 TWICE, the reverse mate no pairing flag at all, arithmetic in uint8."""
    flat = [r for pr in pairs for r in pr]
    res = GswBatchToGiraf(gg, flat, index, seedLen, scoreMatrix, device_seeds=device_seeds, on_panic=on_panic)
    out = []
    for k in range(len(pairs)):
        fwd, rev = res[2 * k], res[2 * k + 1]
        if isinstance(fwd, GoPanic) or isinstance(rev, GoPanic):
            out.append((fwd, rev))
            continue
        fwd.Flag = getGirafFlags(fwd)
        rev.Flag = getGirafFlags(rev)
        fwd.Flag = (fwd.Flag + 8 + 16 + 16) & 0xff
        if isProperPairAlign(fwd, rev):
            fwd.Flag = (fwd.Flag + 1) & 0xff
            rev.Flag = (rev.Flag + 1) & 0xff
        out.append((fwd, rev))
    return out
