"""Test infrastructure: an independent, sequential, literal restatement of the graph aligner's read path
(/root/reference/genomeGraph: index.go:21-121, search.go:135-232 and 338-590, toGiraf.go:17-72, dna/dnaTwoBit/perfectAlign.go,
cigar/tools.go:4-48), with the DPs done by the CPU oracle (oracle/gnx_oracle.c or_gsw_extend).  The product's batched, device-backed
mirror (gonomics_amd/genomeGraph.py) must reproduce it.  Same semantics as the product's parity contract, INCLUDING what Go does through
shared backing arrays (round 4): the route slice that sibling branches of a traversal share, in-place cigar.Append / ReverseCigar, seeds
that point into the re-used nextParts slice -- modelled here with its own Go-slice class (header + shared array + Go 1.25's append
capacities), independently of the product's.  PARITY UNPINNED by the reference."""
import numpy as np

import oracle

U = np.uint64
ONES = U(0xFFFFFFFFFFFFFFFF)


# ---- two-bit words (dnaTwoBit.go:22-37, 66-76; rainbow.go:27-45) ----
def words_of(bases):
    out = []
    b = list(int(x) for x in bases)
    for s in range(0, len(b), 32):
        chunk = b[s:s + 32]
        w = 0
        for x in chunk:
            w = (w * 4 | x) % (1 << 64)
        w = (w * (4 ** (32 - len(chunk)))) % (1 << 64)
        out.append(w)
    return out, len(b)


def rainbow_of(bases):
    return [words_of([0] * o + list(int(x) for x in bases)) for o in range(32)]


def lead0(x):
    n = 0
    for k in range(63, -1, -1):
        if (x >> k) & 1:
            break
        n += 1
    return n


def trail0(x):
    n = 0
    for k in range(64):
        if (x >> k) & 1:
            break
        n += 1
    return n


def count_right(one, start_one, two, start_two):
    (w1, l1), (w2, l2) = one, two
    o1, o2 = (start_one % 32) * 2, (start_two % 32) * 2
    assert o1 == o2
    i, j = start_one // 32, start_two // 32
    d = (w1[i] ^ w2[j]) & (0xFFFFFFFFFFFFFFFF >> o1)
    bm = lead0(d)
    tot = bm - o1
    i += 1
    j += 1
    while i < (l1 + 31) // 32 and j < (l2 + 31) // 32 and bm == 64:
        bm = lead0(w1[i] ^ w2[j])
        tot += bm
        i += 1
        j += 1
    return min(tot // 2, l1 - start_one, l2 - start_two)


def count_left(one, start_one, two, start_two):
    (w1, _), (w2, _) = one, two
    o1 = (start_one % 32) * 2
    assert o1 == (start_two % 32) * 2
    nolook = 64 - o1 - 2
    i, j = start_one // 32, start_two // 32
    d = (w1[i] ^ w2[j]) & ((0xFFFFFFFFFFFFFFFF << nolook) & 0xFFFFFFFFFFFFFFFF)
    bm = trail0(d)
    tot = bm - nolook
    i -= 1
    j -= 1
    while i >= 0 and j >= 0 and bm == 64:
        bm = trail0(w1[i] ^ w2[j])
        tot += bm
        i -= 1
        j -= 1
    return tot // 2


def get_base(tb, pos):
    return (tb[0][pos // 32] >> (64 - 2 * (pos % 32 + 1))) & 3


# ---- graph: nodes = list of dicts {"seq": array, "prev": [ids], "next": [ids]} ----
def make_graph(seqs, edges):
    nodes = [{"id": k, "seq": np.asarray(s, dtype=np.uint8), "prev": [], "next": [], "tb": words_of(s)} for k, s in enumerate(seqs)]
    for u, v in edges:
        nodes[u]["next"].append(v)
        nodes[v]["prev"].append(u)
    return nodes


def to_number(seq):
    a = int(seq[0])
    for x in seq[1:]:
        a = ((a << 2) | int(x)) % (1 << 64)
    return a


def index_genome(nodes, seed_len, seed_step):
    ans = {}

    def helper(prev, nid, loc):
        cur = nodes[nid]
        if len(prev) + len(cur["seq"]) >= seed_len:
            s = prev + [int(x) for x in cur["seq"][:seed_len - len(prev)]]
            if 4 not in s:
                ans.setdefault(to_number(s), []).append(loc)
        else:
            for nx in cur["next"]:
                helper(prev + [int(x) for x in cur["seq"]], nx, loc)

    for nid, n in enumerate(nodes):
        seq = [int(x) for x in n["seq"]]
        pos = 0
        while pos < len(seq) - seed_len + 1:
            if 4 not in seq[pos:pos + seed_len]:
                ans.setdefault(to_number(seq[pos:pos + seed_len]), []).append((nid << 32) | pos)
            pos += seed_step
        while pos < len(seq):
            for nx in n["next"]:
                helper(seq[pos:], nx, (nid << 32) | pos)
            pos += seed_step
    return ans


# ---- seeds: tuples of parts (tid, tstart, qstart, length, pos_strand, total) ----
# Seeds while they are being built: a seed VALUE is (part tuple, pointer to the next part); the pointer is a CELL of the `nextParts`
# array of the call that created it (`NextPart: &nextParts[j]`, search.go:449).  extendToTheRightDev re-uses that array for the next
# edge of node.Next (`nextParts = extendToTheRightDev(..., nextParts)` -> `answer = answer[:0]` -> append in place, search.go:427, 447),
# so a seed made for the first edge can end up pointing at the part made for the second one.  Cells are objects with identity; storing a
# value into a cell keeps the cell.  seed_map flattens every chain at the end, as GraphSmithWatermanToGiraf reads them: after all writes.
class SeedCell:
    __slots__ = ("val",)

    def __init__(self, val=None):
        self.val = val


class SeedSlice:
    __slots__ = ("arr", "n", "cap")

    def __init__(self, arr=None, n=0, cap=0):
        self.arr, self.n, self.cap = arr if arr is not None else [], n, cap


def seed_append(s, val):
    if s.n < s.cap:
        s.arr[s.n].val = val  # in place: whoever points at this cell sees the new seed
        return SeedSlice(s.arr, s.n + 1, s.cap)
    cap = go_next_cap(s.n + 1, s.cap, 32)  # SeedDev: 4 x uint32, bool, uint32, pointer = 32 bytes
    arr = [SeedCell(s.arr[k].val) for k in range(s.n)] + [SeedCell(val)] + [SeedCell() for _ in range(cap - s.n - 1)]
    return SeedSlice(arr, s.n + 1, cap)


def ext_right(nodes, nid, rd, read_start, node_start, pos, answer=None):
    """extendToTheRightDev (search.go:425-461); returns a SeedSlice of values, or None (Go: nil)"""
    answer = SeedSlice(answer.arr, 0, answer.cap) if answer is not None else SeedSlice()
    n = nodes[nid]
    rain = rd["rb"] if pos else rd["rbrc"]
    ro = 31 - ((read_start - node_start % 32 + 31) % 32)
    rm = count_right(n["tb"], node_start, rain[ro], read_start + ro)
    if rm == 0:
        return None
    next_parts = None
    if read_start + rm < len(rd["seq"]) and node_start + rm == n["tb"][1] and n["next"]:
        for nx in n["next"]:
            next_parts = ext_right(nodes, nx, rd, read_start + rm, 0, pos, next_parts)
            for j in range(next_parts.n if next_parts is not None else 0):
                cell = next_parts.arr[j]
                answer = seed_append(answer, ((nid, node_start, read_start, rm, pos, rm + cell.val[0][5]), cell))
    if answer.n == 0:
        answer = SeedSlice([SeedCell(((nid, node_start, read_start, rm, pos, rm), None))], 1, 1)
    return answer


def seed_values(sl):
    return [sl.arr[k].val for k in range(sl.n)] if sl is not None else []


def seed_flat(val):
    """the chain of parts behind a seed value, read NOW"""
    parts, cell = [val[0]], val[1]
    while cell is not None:
        parts.append(cell.val[0]); cell = cell.val[1]
    return tuple(parts)


def left_helper(nodes, nid, rd, nxt):
    """extendToTheLeftHelperDev (search.go:487-531) on seed VALUES; `NextPart: &nextPart` points at the callee's own copy of its argument"""
    n = nodes[nid]
    head = nxt[0]
    pos = head[4]
    rain = rd["rb"] if pos else rd["rbrc"]
    node_pos = n["tb"][1] - 1
    read_pos = head[2] - 1
    ro = 31 - ((read_pos - node_pos % 32 + 31) % 32)
    lm = min(read_pos + 1, count_left(n["tb"], node_pos, rain[ro], read_pos + ro))
    assert lm > 0
    cur = ((nid, node_pos - (lm - 1), read_pos - (lm - 1), lm, pos, lm + head[5]), SeedCell(nxt))
    ans = []
    if cur[0][2] > 0 and cur[0][1] == 0:
        for pv in n["prev"]:
            if get_base(rain[0], cur[0][2] - 1) == get_base(nodes[pv]["tb"], nodes[pv]["tb"][1] - 1):
                ans += left_helper(nodes, pv, rd, cur)
    return ans or [cur]


def ext_left(nodes, nid, rd, cur):
    n = nodes[nid]
    pos = cur[0][4]
    rain = rd["rb"] if pos else rd["rbrc"]
    ans = []
    if cur[0][2] > 0 and cur[0][1] == 0:
        for pv in n["prev"]:
            if get_base(rain[0], cur[0][2] - 1) == get_base(nodes[pv]["tb"], nodes[pv]["tb"][1] - 1):
                ans += left_helper(nodes, pv, rd, cur)
    return ans or [cur]


def heap_sort(a):
    def total(s):
        return s[0][5]

    def heapify(n, i):
        l, r = 2 * i + 1, 2 * i + 2
        mx = l if (l < n and total(a[l]) < total(a[i])) else i
        if r < n and total(a[r]) < total(a[mx]):
            mx = r
        if mx != i:
            a[i], a[mx] = a[mx], a[i]
            heapify(n, mx)

    for i in range(len(a) // 2 - 1, -1, -1):
        heapify(len(a), i)
    size = len(a)
    for i in range(size - 1, 0, -1):
        a[0], a[i] = a[i], a[0]
        size -= 1
        heapify(size, 0)


def make_read(seq):
    seq = np.asarray(seq, dtype=np.uint8)
    rc = np.asarray([3 - int(x) if x < 4 else int(x) for x in seq[::-1]], dtype=np.uint8)
    return {"seq": seq, "rc": rc, "rb": rainbow_of(seq), "rbrc": rainbow_of(rc)}


def seed_map(index, nodes, rd, seed_len, sort=True):
    final = []
    temp = None
    for read_start in range(0, len(rd["seq"]) - seed_len + 1):
        key_idx = (read_start + 31) // 32
        key_off = 31 - ((read_start + 31) % 32)
        for pos, rain in ((True, rd["rb"]), (False, rd["rbrc"])):
            key = rain[key_off][0][key_idx] >> (64 - 2 * seed_len)
            for code in index.get(key, []):
                nid, npos = code >> 32, code & 0xFFFFFFFF
                ro = 31 - ((read_start - npos % 32 + 31) % 32)
                lm = min(read_start + 1, count_left(nodes[nid]["tb"], npos, rain[ro], read_start + ro))
                temp = ext_right(nodes, nid, rd, read_start - (lm - 1), npos - (lm - 1), pos, temp)  # (tempSeeds is re-used from hit to hit: its values are copied out below before the next call)
                if pos:
                    for t in seed_values(temp):
                        final += ext_left(nodes, nid, rd, t)
                else:
                    final += seed_values(temp)
    final = [seed_flat(v) for v in final]  # the chains as GraphSmithWatermanToGiraf will read them: after every write
    if sort:
        if len(final) > 100:
            final.sort(key=lambda s: -s[0][5])  # the documented order (stable); Go: unstable sort.Slice
        else:
            heap_sort(final)
    return final


def could_be_better(seed_len, best, perfect, qlen, mx, mn, lsm, lsc):
    seeds = qlen // (seed_len + 1)
    rem = qlen % (seed_len + 1)
    if seed_len * mx >= best and perfect - ((qlen - seed_len) * mn) >= best:
        return True
    if seed_len * seeds * mx + seeds * lsm >= best and perfect - rem * mn + seeds * lsc >= best:
        return True
    if seed_len * seeds * mx + rem * mx + (seeds + 1) * lsm >= best and perfect + (seeds + 1) * lsc >= best:
        return True
    return False


# ---- Go slices -----------------------------------------------------------------------------------------------------------------
# The traversals hand ONE route slice from sibling branch to sibling branch (search.go:185-195: `dynamicScore.route, ... =
# LeftAlignTraversal(..., dynamicScore, ...)`), keep headers of it (`sk.leftAlignment = dynamicScore.route`), reverse it in place, and
# GraphSmithWatermanToGiraf appends to it in place (cigar.Append / cigar.Concat).  What a later sibling's DP writes through the shared
# backing array is visible through every header that still points into it -- until an append outgrows the capacity and moves on to a
# new array.  A restatement that copies lists ("value semantics") differs from the Go program exactly there (VERDICT r3 missing 1),
# so this one models the slice: (backing array, offset, len, cap) and Go 1.25's growth rule (runtime/slice.go nextslicecap +
# roundupsize over the allocator's size classes; go.mod says go 1.25).  Elements are mutable cells [run, op] standing for the struct values.
GO_SIZE_CLASSES = [0, 8, 16, 24, 32, 48, 64, 80, 96, 112, 128, 144, 160, 176, 192, 208, 224, 240, 256, 288, 320, 352, 384, 416, 448, 480, 512, 576, 640,
                   704, 768, 896, 1024, 1152, 1280, 1408, 1536, 1792, 2048, 2304, 2688, 3072, 3200, 3456, 4096, 4864, 5376, 6144, 6528, 6784, 6912,
                   8192, 9472, 9728, 10240, 10880, 12288, 13568, 14336, 16384, 18432, 19072, 20480, 21760, 24576, 27264, 28672, 32768]


def go_next_cap(new_len, old_cap, elem_size):
    """cap of the array append() allocates when new_len elements no longer fit old_cap"""
    newcap = old_cap
    double = newcap + newcap
    if new_len > double:
        newcap = new_len
    elif old_cap < 256:
        newcap = double
    else:
        while newcap < new_len:
            newcap += (newcap + 3 * 256) >> 2
    mem = newcap * elem_size
    if mem <= 32768:
        mem = next(c for c in GO_SIZE_CLASSES if c >= mem)
    else:
        mem = (mem + 8191) // 8192 * 8192
    return mem // elem_size


class GoSlice:
    """a slice header over a shared backing array (a Python list of cells); copies of the header share the array"""
    __slots__ = ("arr", "off", "n", "cap", "esz")

    def __init__(self, arr=None, off=0, n=0, cap=0, esz=16):
        self.arr, self.off, self.n, self.cap, self.esz = arr if arr is not None else [], off, n, cap, esz

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

    def cells(self):
        return [self.arr[self.off + i] for i in range(self.n)]

    def tail(self, k):  # s[k:]
        return GoSlice(self.arr, self.off + k, self.n - k, self.cap - k, self.esz)


def go_append(s, *vals):
    """append(s, vals...): in place while the capacity lasts, else a new array (old elements copied) of Go's next capacity"""
    need = s.n + len(vals)
    if need <= s.cap:
        for k, v in enumerate(vals):
            s.arr[s.off + s.n + k] = list(v)
        return GoSlice(s.arr, s.off, need, s.cap, s.esz)
    cap = go_next_cap(need, s.cap, s.esz)
    arr = [list(c) for c in s.cells()] + [list(v) for v in vals] + [None] * (cap - need)
    return GoSlice(arr, 0, need, cap, s.esz)


def go_reverse(s):  # cigar.ReverseCigar (cigar/tools.go:43-48): in place
    i, j = 0, s.n - 1
    while i < s.n // 2:
        a, b = s[i], s[j]
        s[i], s[j] = b, a
        i, j = i + 1, j - 1


def dp_merge(route, runs):
    """the route-building loops of LeftDynamicAln / RightDynamicAln (search.go:252-262, 298-308) over the traced steps (`runs`: the
    oracle's run-length form of them, traceback order); resetDynamicScore got its argument by value, so `route` is what came in"""
    idx = 0
    for run, op in runs:
        for _ in range(int(run)):
            if route.n == 0:
                route = go_append(route, [1, op])
            elif route[idx][1] == op:
                route[idx][0] += 1
            else:
                route = go_append(route, [1, op])
                idx += 1
    return route


def _dp(side, scores, target, read, route):
    score, aln, ti, qi = oracle.gsw_extend(side, scores, -600, target, read, route_in=None)  # the DP itself: the pinned C restatement
    return score, dp_merge(route, aln), ti, qi


# ---- traversals (routes as Go slices of cells [run, op 0/1/2]) ----
def left_trav(nodes, nid, seq, ref_end, path, extension, read, scores, route):
    n = nodes[nid]
    # search.go:139 as Go parses it: ((refEnd - Min(len(seq)+refEnd, extension)) - len(seq)); a negative bound is a Go panic
    lo = ref_end - min(len(seq) + ref_end, extension)
    lo = lo - len(seq)
    if lo < 0:
        raise IndexError("slice bounds out of range [%d:%d]" % (lo, ref_end))
    s_seq = [int(x) for x in n["seq"][lo:ref_end]] + list(seq)
    s_path = list(path)  # AddPath's result is dropped (search.go:176): the node is not recorded (all paths stay empty: nothing to alias)
    if len(seq) + ref_end >= extension or not n["prev"]:
        score, aln, ti, qi = _dp(0, scores, s_seq, read, route)
        return aln, score, ref_end - len(s_seq) - len(seq) + ti, qi, s_path
    best, best_score = None, -(1 << 63)
    for pv in n["prev"]:
        route, cs, ts, qs, cp = left_trav(nodes, pv, s_seq, len(nodes[pv]["seq"]), s_path, extension, read, scores, route)
        s_path = cp  # (`s.Path` is assigned the sibling's path and handed to the next one)
        if cs > best_score:
            best_score = cs
            best = (route, ref_end - len(s_seq) - len(seq) + ts, qs, cp)  # the HEADER: later siblings write through the same array
    aln, ts, qs, cp = best
    go_reverse(aln)
    return aln, best_score, ts, qs, cp[::-1]


def right_trav(nodes, nid, seq, start, path, extension, read, scores, route):
    n = nodes[nid]
    take = min(len(seq) + len(n["seq"]) - start, extension) - len(seq)
    s_seq = list(seq) + [int(x) for x in n["seq"][start:start + take]]
    s_path = list(path)
    if len(seq) + len(n["seq"]) - start >= extension or not n["next"]:
        score, aln, te, qe = _dp(1, scores, s_seq, read, route)
        return aln, score, te + start, qe, s_path
    best, best_score = None, -(1 << 63)
    for nx in n["next"]:
        route, cs, te, qe, cp = right_trav(nodes, nx, s_seq, 0, s_path, extension, read, scores, route)
        s_path = cp
        if cs > best_score:
            best_score = cs
            best = (route, te, qe, cp)
    aln, te, qe, cp = best
    go_reverse(aln)
    return aln, best_score, te + start, qe, cp


LETTER = {0: ord("M"), 1: ord("I"), 2: ord("D")}
CONSUMES_QUERY = (ord("M"), ord("I"), ord("S"), ord("="), ord("X"))


def cat_paths(a, b):
    if not b:
        return a
    if not a:
        return b
    a = list(a)
    if a[-1] != b[0]:
        a.append(b[0])
    return a + list(b[1:])


def cig_append(alpha, beta):  # cigar.Append (cigar/tools.go:4-11)
    if alpha.n > 0 and alpha[alpha.n - 1][1] == beta[1]:
        alpha[alpha.n - 1][0] += beta[0]
    else:
        alpha = go_append(alpha, beta)
    return alpha


def cig_concat(alpha, beta):  # cigar.Concat (cigar/tools.go:14-23)
    if alpha.n == 0:
        return beta
    if beta.n > 0:
        alpha = cig_append(alpha, beta[0])
        beta = beta.tail(1)
    return go_append(alpha, *beta.cells())


def soft_clips(front, length, cigs):  # cigar.AppendSoftClips (cigar/tools.go:26-40)
    run = sum(c[0] for c in cigs.cells() if c[1] in CONSUMES_QUERY)
    if front == 0 and run >= length:
        return cigs
    ans = GoSlice(arr=[None] * (cigs.n + 2), off=0, n=0, cap=cigs.n + 2)
    if front > 0:
        ans = go_append(ans, [front, ord("S")])
    if front + run < length:
        ans = go_append(go_append(ans, *cigs.cells()), [length - front - run, ord("S")])
    return ans


def read_to_giraf(nodes, rd, seeds, scores):
    """GraphSmithWatermanToGiraf (toGiraf.go:17-72).  The routes come back with op codes 0/1/2 (the oracle's) and are turned into the
    letters 'M' 'I' 'D' cell by cell IN PLACE before cigar.Append sees them, which keeps every aliasing relation (the Go routes hold
    the letters from the start)."""
    sc = np.asarray(scores, dtype=np.int64)
    seq = rd["seq"]
    best = {"QStart": 0, "QEnd": 0, "PosStrand": True, "Path": (0, [], 0), "Cigar": None, "AlnScore": 0, "Seq": seq}
    perfect = int(sum(sc[int(x)][int(x)] for x in seq))
    extension = perfect // 600 + len(seq)
    left_aln, right_aln, left_path, right_path, q_end = GoSlice(), GoSlice(), [], [], 0  # sk.* survive from seed to seed (resetScoreKeeper: by value)
    for seed in seeds:
        head, tail = seed[0], seed[-1]
        if not could_be_better(head[5], best["AlnScore"], perfect, len(seq), 100, 90, -196, -296):
            break
        cur = seq if head[4] else rd["rc"]
        seed_score = int(sum(sc[int(x)][int(x)] for x in cur[head[2]:tail[2] + tail[3]]))
        if head[5] == len(cur):
            t_start, t_end, q_start, score = head[1], tail[1] + tail[3], head[2], seed_score
        else:
            ext = extension - head[5]
            left_aln, ls, t_start, q_start, left_path = left_trav(nodes, head[0], [], head[1], left_path, ext, [int(x) for x in cur[:head[2]]], scores, GoSlice())
            right_aln, rs, t_end, q_end, right_path = right_trav(nodes, tail[0], [], tail[1] + tail[3], right_path, ext, [int(x) for x in cur[tail[2] + tail[3]:]], scores, GoSlice())
            for c in left_aln.cells() + right_aln.cells():  # op codes -> letters, once per cell (a cell reached through both slices: once)
                if c[1] in LETTER:
                    c[1] = LETTER[c[1]]
            score = ls + seed_score + rs
        if score > best["AlnScore"]:
            cig = soft_clips(q_start, len(cur), cig_concat(cig_append(left_aln, [head[5], ord("M")]), right_aln))
            best = {"QStart": q_start, "QEnd": head[2] + q_start + q_end + head[5] - 1, "PosStrand": head[4],
                    "Path": (t_start, cat_paths(cat_paths(left_path, [p[0] for p in seed]), right_path), t_end),
                    "Cigar": cig, "AlnScore": score, "Seq": cur}
    if best["Cigar"] is not None:
        best["Cigar"] = [tuple(c) for c in best["Cigar"].cells()]  # what the caller sees when the function returns
    return best


def giraf_key(g):
    return (g["QStart"], g["QEnd"], g["PosStrand"], g["Path"][0], tuple(g["Path"][1]), g["Path"][2],
            None if g["Cigar"] is None else tuple(g["Cigar"]), g["AlnScore"], bytes(np.asarray(g["Seq"], dtype=np.uint8)))


def wrap_pair(nodes, rd_fwd, rd_rev, seeds_fwd, seeds_rev, scores):
    """WrapPairGiraf + setGirafFlags (toGiraf.go:117-140), restated from the Go text: flags are uint8; Fwd gets 8, 16 and 16 again."""
    f = read_to_giraf(nodes, rd_fwd, seeds_fwd, scores)
    r = read_to_giraf(nodes, rd_rev, seeds_rev, scores)
    def flags(g):
        a = 0
        if g["PosStrand"]:
            a += 4
        if g["AlnScore"] < 1200:
            a += 2
        return a
    ff, rf = flags(f), flags(r)
    ff += 8
    ff += 16
    ff += 16
    proper = False
    if abs(float(f["Path"][0] - r["Path"][0])) < 10000:
        if f["Path"][0] < r["Path"][0] and f["PosStrand"] and not r["PosStrand"]:
            proper = True
        if f["Path"][0] > r["Path"][0] and (not f["PosStrand"]) and r["PosStrand"]:
            proper = True
    if proper:
        ff += 1
        rf += 1
    return f, r, ff % 256, rf % 256
