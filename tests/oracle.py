"""ctypes wrapper of oracle/liboracle.so (the CPU restatement of the reference path) -- test infrastructure.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")

MODE_AFFINE, MODE_CONST, MODE_AFFINE_HIGHMEM, MODE_AFFINE_LOCAL, MODE_CONST_HIGHMEM = range(5)

CIGAR_DTYPE = np.dtype({"names": ["run_length", "op"], "formats": [np.int64, np.uint8], "offsets": [0, 8], "itemsize": 16})

_lib = None


def build():
    src = os.path.join(ORACLE_DIR, "gnx_oracle.c")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(LIB)
        c_p, i64 = ctypes.c_void_p, ctypes.c_int64
        L.or_align_batch.argtypes = [ctypes.c_int, c_p, i64, i64, i64, i64, i64, c_p, c_p, c_p, c_p, ctypes.c_int, c_p,
                                     ctypes.POINTER(c_p), c_p]
        L.or_align_batch.restype = ctypes.c_int
        L.or_affine_gap_chunk.argtypes = [c_p, i64, c_p, i64, c_p, i64, i64, i64, ctypes.POINTER(i64), ctypes.POINTER(c_p), ctypes.POINTER(i64)]
        L.or_affine_gap_chunk.restype = ctypes.c_int
        L.or_multiple_affine_gap.argtypes = [c_p, ctypes.c_int, i64, c_p, ctypes.c_int, i64, c_p, i64, i64, i64,
                                             ctypes.POINTER(i64), ctypes.POINTER(c_p), ctypes.POINTER(i64)]
        L.or_multiple_affine_gap.restype = ctypes.c_int
        L.or_gsw_extend.argtypes = [ctypes.c_int, c_p, i64, c_p, i64, c_p, i64, c_p, i64, i64, ctypes.POINTER(i64), ctypes.POINTER(i64),
                                    ctypes.POINTER(i64), ctypes.POINTER(c_p), ctypes.POINTER(i64)]
        L.or_gsw_extend.restype = ctypes.c_int
        L.or_free.argtypes = [c_p]
        L.or_free.restype = None
        _lib = L
    return _lib


class OracleError(RuntimeError):
    pass


def align_batch_windows(mode, scores, gap_open, gap_extend, a_buf, a_start, a_len, b_buf, b_start, b_len,
                        ci=10000, cj=10000, threads=1):
    """Runs the oracle on windows of shared buffers (copies them into the concatenated layout it expects)."""
    a_buf = np.ascontiguousarray(a_buf, dtype=np.uint8)
    b_buf = np.ascontiguousarray(b_buf, dtype=np.uint8)
    alphas = [a_buf[s:s + l] for s, l in zip(a_start, a_len)]
    betas = [b_buf[s:s + l] for s, l in zip(b_start, b_len)]
    return align_batch(mode, scores, gap_open, gap_extend, alphas, betas, ci, cj, threads)


def align_batch(mode, scores, gap_open, gap_extend, alphas, betas, ci=10000, cj=10000, threads=1):
    """Returns (scores[int64], ops[CIGAR_DTYPE], off[int64 n+1])."""
    L = lib()
    n = len(alphas)
    sc = np.ascontiguousarray(np.asarray(scores, dtype=np.int64).reshape(25))
    a_off = np.zeros(n + 1, dtype=np.int64)
    b_off = np.zeros(n + 1, dtype=np.int64)
    if n:
        a_off[1:] = np.cumsum([len(a) for a in alphas])
        b_off[1:] = np.cumsum([len(b) for b in betas])
    a_cat = np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.uint8) for a in alphas] + [np.zeros(1, np.uint8)]))
    b_cat = np.ascontiguousarray(np.concatenate([np.asarray(b, dtype=np.uint8) for b in betas] + [np.zeros(1, np.uint8)]))
    out_score = np.zeros(max(n, 1), dtype=np.int64)
    out_off = np.zeros(n + 1, dtype=np.int64)
    ops_p = ctypes.c_void_p()
    rc = L.or_align_batch(mode, sc.ctypes.data, int(gap_open), int(gap_extend), int(ci), int(cj), n,
                          a_cat.ctypes.data, a_off.ctypes.data, b_cat.ctypes.data, b_off.ctypes.data, int(threads),
                          out_score.ctypes.data, ctypes.byref(ops_p), out_off.ctypes.data)
    if rc != 0:
        raise OracleError("oracle error %d" % rc)
    total = int(out_off[-1])
    if total:
        buf = (ctypes.c_char * (total * 16)).from_address(ops_p.value)
        ops = np.frombuffer(buf, dtype=CIGAR_DTYPE, count=total).copy()
    else:
        ops = np.zeros(0, dtype=CIGAR_DTYPE)
    L.or_free(ops_p)
    return out_score[:n], ops, out_off


def align_one(mode, scores, gap_open, gap_extend, alpha, beta, ci=10000, cj=10000):
    s, ops, off = align_batch(mode, scores, gap_open, gap_extend, [alpha], [beta], ci, cj)
    return int(s[0]), [(int(r), int(o)) for r, o in zip(ops["run_length"], ops["op"])]


def cigar_str(route):
    return "".join("%d%s" % (r, "MID"[o]) for r, o in route)


def _route_out(rc, score, ops_p, nops):
    if rc != 0:
        raise OracleError("oracle error %d" % rc)
    total = nops.value
    buf = (ctypes.c_char * (max(total, 1) * 16)).from_address(ops_p.value)
    ops = np.frombuffer(buf, dtype=CIGAR_DTYPE, count=total).copy()
    lib().or_free(ops_p)
    return int(score.value), [(int(r), int(o)) for r, o in zip(ops["run_length"], ops["op"])]


def affine_gap_chunk(scores, gap_open, gap_extend, chunk, alpha, beta):
    L = lib()
    sc = np.ascontiguousarray(np.asarray(scores, dtype=np.int64).reshape(25))
    a = np.ascontiguousarray(np.concatenate([np.asarray(alpha, np.uint8), np.zeros(1, np.uint8)]))
    b = np.ascontiguousarray(np.concatenate([np.asarray(beta, np.uint8), np.zeros(1, np.uint8)]))
    score, nops, ops_p = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_void_p()
    rc = L.or_affine_gap_chunk(a.ctypes.data, len(alpha), b.ctypes.data, len(beta), sc.ctypes.data, int(gap_open), int(gap_extend), int(chunk),
                               ctypes.byref(score), ctypes.byref(ops_p), ctypes.byref(nops))
    return _route_out(rc, score, ops_p, nops)


def multiple_affine_gap(scores, gap_open, gap_extend, chunk, block_a, block_b):
    """block_* : 2-D uint8 arrays (nseq x len)."""
    L = lib()
    sc = np.ascontiguousarray(np.asarray(scores, dtype=np.int64).reshape(25))
    A = np.ascontiguousarray(block_a, dtype=np.uint8)
    B = np.ascontiguousarray(block_b, dtype=np.uint8)
    Ab = np.ascontiguousarray(np.concatenate([A.reshape(-1), np.zeros(1, np.uint8)]))
    Bb = np.ascontiguousarray(np.concatenate([B.reshape(-1), np.zeros(1, np.uint8)]))
    score, nops, ops_p = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_void_p()
    rc = L.or_multiple_affine_gap(Ab.ctypes.data, A.shape[0], A.shape[1], Bb.ctypes.data, B.shape[0], B.shape[1], sc.ctypes.data,
                                  int(gap_open), int(gap_extend), int(chunk), ctypes.byref(score), ctypes.byref(ops_p), ctypes.byref(nops))
    return _route_out(rc, score, ops_p, nops)


def gsw_extend(side, scores, gap_pen, alpha, beta, route_in=None, curr_max=0):
    """genomeGraph.LeftDynamicAln (side 0) / RightDynamicAln (side 1) restated in oracle/gnx_oracle.c.
    route_in: [(run, op)] with ops 0/1/2 (the caller's dynamicScore.route, kept because resetDynamicScore is a no-op).
    Returns (score, route [(run, op)], i, j)."""
    L = lib()
    sc = np.ascontiguousarray(np.asarray(scores, dtype=np.int64).reshape(25))
    a = np.ascontiguousarray(alpha, dtype=np.uint8)
    b = np.ascontiguousarray(beta, dtype=np.uint8)
    rin = np.zeros(max(len(route_in or []), 1), dtype=CIGAR_DTYPE)
    for k, (r, o) in enumerate(route_in or []):
        rin[k] = (r, o)
    score, oi, oj, n_out = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    out_p = ctypes.c_void_p()
    rc = L.or_gsw_extend(int(side), a.ctypes.data if a.size else None, a.shape[0], b.ctypes.data if b.size else None, b.shape[0], sc.ctypes.data,
                         int(gap_pen), rin.ctypes.data, len(route_in or []), int(curr_max), ctypes.byref(score), ctypes.byref(oi), ctypes.byref(oj),
                         ctypes.byref(out_p), ctypes.byref(n_out))
    if rc:
        raise OracleError("or_gsw_extend rc=%d" % rc)
    n = n_out.value
    route = []
    if n:
        buf = (ctypes.c_char * (n * 16)).from_address(out_p.value)
        arr = np.frombuffer(buf, dtype=CIGAR_DTYPE, count=n)
        route = [(int(arr["run_length"][k]), int(arr["op"][k])) for k in range(n)]
    L.or_free(out_p)
    return score.value, route, oi.value, oj.value
