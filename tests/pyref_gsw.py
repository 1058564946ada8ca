"""Test infrastructure: an independent, sequential, literal restatement of the graph aligner's read path
(/root/reference/genomeGraph: index.go:21-121, search.go:135-232 and 338-590, toGiraf.go:17-72, dna/dnaTwoBit/perfectAlign.go,
cigar/tools.go:4-48), with the DPs done by the CPU oracle (oracle/gnx_oracle.c or_gsw_extend).  The product's batched, device-backed
mirror (gonomics_amd/genomeGraph.py) must reproduce it.  Same semantics as the product's parity contract: value semantics where
Go aliases backing arrays across sibling branches; the route carry-over between siblings is kept.  PARITY UNPINNED by the reference."""
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
def ext_right(nodes, nid, rd, read_start, node_start, pos):
    n = nodes[nid]
    rain = rd["rb"] if pos else rd["rbrc"]
    ro = 31 - ((read_start - node_start % 32 + 31) % 32)
    rm = count_right(n["tb"], node_start, rain[ro], read_start + ro)
    if rm == 0:
        return []
    ans = []
    if read_start + rm < len(rd["seq"]) and node_start + rm == n["tb"][1] and n["next"]:
        for nx in n["next"]:
            for parts in ext_right(nodes, nx, rd, read_start + rm, 0, pos):
                ans.append(((nid, node_start, read_start, rm, pos, rm + parts[0][5]),) + parts)
    if not ans:
        ans = [((nid, node_start, read_start, rm, pos, rm),)]
    return ans


def left_helper(nodes, nid, rd, nxt):
    n = nodes[nid]
    head = nxt[0]
    pos = head[4]
    rain = rd["rb"] if pos else rd["rbrc"]
    node_pos = n["tb"][1] - 1
    read_pos = head[2] - 1
    ro = 31 - ((read_pos - node_pos % 32 + 31) % 32)
    lm = min(read_pos + 1, count_left(n["tb"], node_pos, rain[ro], read_pos + ro))
    assert lm > 0
    cur = ((nid, node_pos - (lm - 1), read_pos - (lm - 1), lm, pos, lm + head[5]),) + nxt
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
    for read_start in range(0, len(rd["seq"]) - seed_len + 1):
        key_idx = (read_start + 31) // 32
        key_off = 31 - ((read_start + 31) % 32)
        for pos, rain in ((True, rd["rb"]), (False, rd["rbrc"])):
            key = rain[key_off][0][key_idx] >> (64 - 2 * seed_len)
            for code in index.get(key, []):
                nid, npos = code >> 32, code & 0xFFFFFFFF
                ro = 31 - ((read_start - npos % 32 + 31) % 32)
                lm = min(read_start + 1, count_left(nodes[nid]["tb"], npos, rain[ro], read_start + ro))
                temp = ext_right(nodes, nid, rd, read_start - (lm - 1), npos - (lm - 1), pos)
                if pos:
                    for t in temp:
                        final += ext_left(nodes, nid, rd, t)
                else:
                    final += temp
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


# ---- traversals (routes as [(run, op 0/1/2)], the oracle's format) ----
def left_trav(nodes, nid, seq, ref_end, path, extension, read, scores, route):
    n = nodes[nid]
    # search.go:139 as Go parses it: ((refEnd - Min(len(seq)+refEnd, extension)) - len(seq)); a negative bound is a Go panic
    lo = ref_end - min(len(seq) + ref_end, extension)
    lo = lo - len(seq)
    if lo < 0:
        raise IndexError("slice bounds out of range [%d:%d]" % (lo, ref_end))
    s_seq = [int(x) for x in n["seq"][lo:ref_end]] + list(seq)
    s_path = list(path)  # AddPath's result is dropped (search.go:176): the node is not recorded
    if len(seq) + ref_end >= extension or not n["prev"]:
        score, aln, ti, qi = oracle.gsw_extend(0, scores, -600, s_seq, read, route_in=route)
        return aln, score, ref_end - len(s_seq) - len(seq) + ti, qi, s_path
    best, best_score = None, -(1 << 63)
    for pv in n["prev"]:
        route, cs, ts, qs, cp = left_trav(nodes, pv, s_seq, len(nodes[pv]["seq"]), s_path, extension, read, scores, route)
        if cs > best_score:
            best_score = cs
            best = (list(route), ref_end - len(s_seq) - len(seq) + ts, qs, list(cp))
    aln, ts, qs, cp = best
    return aln[::-1], best_score, ts, qs, cp[::-1]


def right_trav(nodes, nid, seq, start, path, extension, read, scores, route):
    n = nodes[nid]
    take = min(len(seq) + len(n["seq"]) - start, extension) - len(seq)
    s_seq = list(seq) + [int(x) for x in n["seq"][start:start + take]]
    s_path = list(path)
    if len(seq) + len(n["seq"]) - start >= extension or not n["next"]:
        score, aln, te, qe = oracle.gsw_extend(1, scores, -600, s_seq, read, route_in=route)
        return aln, score, te + start, qe, s_path
    best, best_score = None, -(1 << 63)
    for nx in n["next"]:
        route, cs, te, qe, cp = right_trav(nodes, nx, s_seq, 0, s_path, extension, read, scores, route)
        if cs > best_score:
            best_score = cs
            best = (list(route), te, qe, list(cp))
    aln, te, qe, cp = best
    return aln[::-1], best_score, te + start, qe, cp


LETTER = {0: ord("M"), 1: ord("I"), 2: ord("D")}


def soft_clips(front, length, cigs):
    run = sum(r for r, o in cigs if o in (ord("M"), ord("I"), ord("S"), ord("="), ord("X")))
    if front == 0 and run >= length:
        return cigs
    ans = []
    if front > 0:
        ans.append((front, ord("S")))
    if front + run < length:
        ans = ans + list(cigs) + [(length - front - run, ord("S"))]
    return ans


def cat_paths(a, b):
    if not b:
        return a
    if not a:
        return b
    a = list(a)
    if a[-1] != b[0]:
        a.append(b[0])
    return a + list(b[1:])


def cig_append(a, x):
    a = list(a)
    if a and a[-1][1] == x[1]:
        a[-1] = (a[-1][0] + x[0], x[1])
    else:
        a.append(x)
    return a


def cig_concat(a, b):
    if not a:
        return list(b)
    if b:
        a = cig_append(a, b[0])
        b = b[1:]
    return list(a) + list(b)


def read_to_giraf(nodes, rd, seeds, scores):
    sc = np.asarray(scores, dtype=np.int64)
    seq = rd["seq"]
    best = {"QStart": 0, "QEnd": 0, "PosStrand": True, "Path": (0, [], 0), "Cigar": None, "AlnScore": 0, "Seq": seq}
    perfect = int(sum(sc[int(x)][int(x)] for x in seq))
    extension = perfect // 600 + len(seq)
    left_aln, right_aln, left_path, right_path, q_end = [], [], [], [], 0
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
            left_aln, ls, t_start, q_start, left_path = left_trav(nodes, head[0], [], head[1], [], ext, [int(x) for x in cur[:head[2]]], scores, None)
            right_aln, rs, t_end, q_end, right_path = right_trav(nodes, tail[0], [], tail[1] + tail[3], [], ext, [int(x) for x in cur[tail[2] + tail[3]:]], scores, None)
            score = ls + seed_score + rs
        if score > best["AlnScore"]:
            la = [(r, LETTER[o]) for r, o in left_aln]
            ra = [(r, LETTER[o]) for r, o in right_aln]
            best = {"QStart": q_start, "QEnd": head[2] + q_start + q_end + head[5] - 1, "PosStrand": head[4],
                    "Path": (t_start, cat_paths(cat_paths(left_path, [p[0] for p in seed]), right_path), t_end),
                    "Cigar": soft_clips(q_start, len(cur), cig_concat(cig_append(la, (head[5], ord("M"))), ra)), "AlnScore": score, "Seq": cur}
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
