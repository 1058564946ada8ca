"""ctypes binding of the C ABI declared in include/gnx_align.h.

The HIP library is the product: if libgonomics_align_hip.so is missing or a GPU is absent, calls fail
loudly (GnxError / OSError).  There is no CPU fallback and nothing here ever touches oracle/.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GNX_LIB_PATH") or os.path.join(_HERE, "libgonomics_align_hip.so")

GNX_OK, GNX_EINVAL, GNX_EBASE, GNX_EEMPTY, GNX_ERANGE, GNX_EDEVICE, GNX_ENOMEM, GNX_ECAPACITY, GNX_ETRACE, GNX_EDIVZERO, GNX_ESTALE = range(11)
GNX_AFFINE_GAP, GNX_CONST_GAP, GNX_AFFINE_GAP_HIGHMEM, GNX_AFFINE_GAP_LOCAL, GNX_CONST_GAP_HIGHMEM = range(5)

EXPORTS = ["gnx_device_count", "gnx_init", "gnx_shutdown", "gnx_last_error", "gnx_free", "gnx_align_batch",
           "gnx_align_batch_windows", "gnx_align_pair", "gnx_align_batch_device", "gnx_get_timing",
           "gnx_affine_gap_chunk_batch", "gnx_multiple_affine_gap_batch", "gnx_gsw_extend_batch",
           "gnx_init_devices", "gnx_n_devices", "gnx_set_reference", "gnx_set_reference_synthetic", "gnx_align_batch_by_offset",
           "gnx_seed_index_build", "gnx_seed_index_set", "gnx_seed_find_batch", "gnx_seed_index_set_gen", "gnx_seed_find_batch_gen", "gnx_gsw_graph_create", "gnx_gsw_graph_free", "gnx_gsw_map_reads", "gnx_debug_occupy", "gnx_debug_counter", "gnx_reference_info"]


class GnxCigar(ctypes.Structure):
    _fields_ = [("run_length", ctypes.c_int64), ("op", ctypes.c_uint8), ("_pad", ctypes.c_uint8 * 7)]


CIGAR_DTYPE = np.dtype({"names": ["run_length", "op"], "formats": [np.int64, np.uint8], "offsets": [0, 8], "itemsize": 16})


class GnxParams(ctypes.Structure):
    _fields_ = [("mode", ctypes.c_int32), ("_reserved", ctypes.c_int32), ("scores", ctypes.c_int64 * 25),
                ("gap_open", ctypes.c_int64), ("gap_extend", ctypes.c_int64),
                ("checkersize_i", ctypes.c_int64), ("checkersize_j", ctypes.c_int64)]


class GnxTiming(ctypes.Structure):
    _fields_ = [("fill_ms", ctypes.c_double), ("traceback_ms", ctypes.c_double), ("total_ms", ctypes.c_double),
                ("cells", ctypes.c_int64), ("n_launches", ctypes.c_int64), ("trace_bytes", ctypes.c_int64),
                ("dominant_ms", ctypes.c_double), ("dominant_launches", ctypes.c_int64), ("fast_path", ctypes.c_int32),
                ("_pad", ctypes.c_int32), ("host_ms", ctypes.c_double), ("stage0_ms", ctypes.c_double), ("fetch_ms", ctypes.c_double),
                ("transport", ctypes.c_int32), ("n_contexts", ctypes.c_int32), ("gather_ms", ctypes.c_double), ("bcast_ms", ctypes.c_double)]


class GnxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("gnx error %d: %s" % (code, msg))
        self.code = code


_lib = None


def lib():
    """Load the HIP library (once).  Raises OSError if it was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        c_p = ctypes.c_void_p
        i64 = ctypes.c_int64
        L.gnx_device_count.restype = ctypes.c_int
        L.gnx_init.argtypes = [ctypes.c_int, i64]
        L.gnx_init.restype = ctypes.c_int
        L.gnx_shutdown.restype = None
        L.gnx_last_error.restype = ctypes.c_char_p
        L.gnx_free.argtypes = [c_p]
        L.gnx_free.restype = None
        L.gnx_align_batch.argtypes = [ctypes.POINTER(GnxParams), i64, c_p, c_p, c_p, c_p, c_p,
                                      ctypes.POINTER(c_p), ctypes.POINTER(c_p)]
        L.gnx_align_batch.restype = ctypes.c_int
        L.gnx_align_batch_windows.argtypes = [ctypes.POINTER(GnxParams), i64, c_p, i64, c_p, c_p, c_p, i64, c_p, c_p, c_p,
                                              ctypes.POINTER(c_p), ctypes.POINTER(c_p)]
        L.gnx_align_batch_windows.restype = ctypes.c_int
        L.gnx_align_pair.argtypes = [ctypes.POINTER(GnxParams), c_p, i64, c_p, i64, ctypes.POINTER(i64),
                                     ctypes.POINTER(c_p), ctypes.POINTER(i64)]
        L.gnx_align_pair.restype = ctypes.c_int
        L.gnx_align_batch_device.argtypes = [ctypes.POINTER(GnxParams), i64, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                             c_p, c_p, i64, c_p, ctypes.POINTER(i64), c_p]
        L.gnx_align_batch_device.restype = ctypes.c_int
        L.gnx_affine_gap_chunk_batch.argtypes = [ctypes.POINTER(GnxParams), i64, i64, c_p, c_p, c_p, c_p, c_p,
                                                 ctypes.POINTER(c_p), ctypes.POINTER(c_p)]
        L.gnx_affine_gap_chunk_batch.restype = ctypes.c_int
        L.gnx_multiple_affine_gap_batch.argtypes = [ctypes.POINTER(GnxParams), i64, i64, c_p, c_p, c_p, c_p, i64, c_p, c_p, c_p,
                                                    ctypes.POINTER(c_p), ctypes.POINTER(c_p)]
        L.gnx_multiple_affine_gap_batch.restype = ctypes.c_int
        L.gnx_gsw_extend_batch.argtypes = [ctypes.c_int, c_p, i64, i64, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                           ctypes.POINTER(c_p), ctypes.POINTER(c_p)]
        L.gnx_gsw_extend_batch.restype = ctypes.c_int
        L.gnx_init_devices.argtypes = [ctypes.c_int, c_p, i64]
        L.gnx_init_devices.restype = ctypes.c_int
        L.gnx_n_devices.restype = ctypes.c_int
        L.gnx_set_reference.argtypes = [c_p, i64]
        L.gnx_set_reference.restype = ctypes.c_int
        L.gnx_set_reference_synthetic.argtypes = [i64, ctypes.c_uint64]
        L.gnx_set_reference_synthetic.restype = ctypes.c_int
        L.gnx_align_batch_by_offset.argtypes = [ctypes.POINTER(GnxParams), i64, c_p, c_p, c_p, c_p, c_p, ctypes.POINTER(c_p), ctypes.POINTER(c_p)]
        L.gnx_align_batch_by_offset.restype = ctypes.c_int
        L.gnx_seed_index_build.argtypes = [c_p, c_p, i64, ctypes.c_int, ctypes.c_int, ctypes.POINTER(c_p), ctypes.POINTER(c_p), ctypes.POINTER(i64)]
        L.gnx_seed_index_build.restype = ctypes.c_int
        L.gnx_seed_index_set.argtypes = [c_p, c_p, i64, c_p, c_p, i64, ctypes.c_int]
        L.gnx_seed_index_set.restype = ctypes.c_int
        L.gnx_seed_find_batch.argtypes = [c_p, c_p, i64, ctypes.POINTER(c_p), ctypes.POINTER(c_p)]
        L.gnx_seed_find_batch.restype = ctypes.c_int
        if hasattr(L, "gnx_gsw_map_reads"):
            L.gnx_gsw_graph_create.argtypes = [c_p, c_p, i64, c_p, c_p, i64, ctypes.c_int, ctypes.c_int, ctypes.POINTER(c_p)]
            L.gnx_gsw_graph_create.restype = ctypes.c_int
            L.gnx_gsw_graph_free.argtypes = [c_p]
            L.gnx_gsw_graph_free.restype = None
            L.gnx_gsw_map_reads.argtypes = [c_p, c_p, c_p, i64, ctypes.c_int, c_p, i64, ctypes.c_int, ctypes.POINTER(c_p), ctypes.POINTER(c_p), ctypes.POINTER(c_p)]
            L.gnx_gsw_map_reads.restype = ctypes.c_int
        L.gnx_get_timing.argtypes = [ctypes.POINTER(GnxTiming)]
        L.gnx_get_timing.restype = ctypes.c_int
        if hasattr(L, "gnx_debug_occupy"):  # (absent from older builds loaded through GNX_LIB_PATH for A/B runs)
            L.gnx_debug_occupy.argtypes = [ctypes.c_int, ctypes.c_int]
            L.gnx_debug_occupy.restype = ctypes.c_int
        if hasattr(L, "gnx_debug_counter"):
            L.gnx_debug_counter.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int64)]
            L.gnx_debug_counter.restype = ctypes.c_int
        _lib = L
    return _lib


def debug_counter(which=0, reset=True):
    v = ctypes.c_int64()
    check(lib().gnx_debug_counter(which, 1 if reset else 0, ctypes.byref(v)))
    return int(v.value)


def check(rc):
    if rc != GNX_OK:
        raise GnxError(rc, lib().gnx_last_error().decode("utf-8", "replace"))


def make_params(mode, scores, gap_open, gap_extend=0, checkersize_i=10000, checkersize_j=10000):
    p = GnxParams()
    p.mode = mode
    flat = [int(v) for row in scores for v in row]
    if len(flat) != 25:
        raise ValueError("scores must be a 5x5 matrix")
    for k, v in enumerate(flat):
        p.scores[k] = v
    p.gap_open = int(gap_open)
    p.gap_extend = int(gap_extend)
    p.checkersize_i = int(checkersize_i)
    p.checkersize_j = int(checkersize_j)
    return p


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _take(ops_p, off_p, n_pairs):
    L = lib()
    off = np.ctypeslib.as_array(ctypes.cast(off_p, ctypes.POINTER(ctypes.c_int64)), shape=(n_pairs + 1,)).copy()
    total = int(off[-1])
    if total:
        buf = (ctypes.c_char * (total * 16)).from_address(ops_p.value)
        ops = np.frombuffer(buf, dtype=CIGAR_DTYPE, count=total).copy()
    else:
        ops = np.zeros(0, dtype=CIGAR_DTYPE)
    L.gnx_free(ops_p)
    L.gnx_free(off_p)
    return ops, off


def align_batch_windows(params, a_buf, a_start, a_len, b_buf, b_start, b_len):
    """Host-buffer batch over windows of shared buffers.  Returns (scores[int64], ops[CIGAR_DTYPE], off[int64])."""
    L = lib()
    a_buf, b_buf = _u8(a_buf), _u8(b_buf)
    a_start, a_len, b_start, b_len = _i64(a_start), _i64(a_len), _i64(b_start), _i64(b_len)
    n = int(a_start.shape[0])
    scores = np.zeros(max(n, 1), dtype=np.int64)
    ops_p, off_p = ctypes.c_void_p(), ctypes.c_void_p()
    check(L.gnx_align_batch_windows(ctypes.byref(params), n, a_buf.ctypes.data, a_buf.shape[0], a_start.ctypes.data, a_len.ctypes.data,
                                    b_buf.ctypes.data, b_buf.shape[0], b_start.ctypes.data, b_len.ctypes.data,
                                    scores.ctypes.data, ctypes.byref(ops_p), ctypes.byref(off_p)))
    ops, off = _take(ops_p, off_p, n)
    return scores[:n], ops, off


class RawResult:
    """Results of a host-buffer call as the C ABI hands them over: `scores` is the caller's array, `ops` / `off` are VIEWS of the
    pinned arrays the library returned (no copy -- what a cgo shim wraps with unsafe.Slice); free() gives them back (gnx_free)."""

    def __init__(self, scores, ops_p, off_p, n):
        self.scores, self._ops_p, self._off_p, self.n = scores, ops_p, off_p, n
        self.off = np.ctypeslib.as_array(ctypes.cast(off_p, ctypes.POINTER(ctypes.c_int64)), shape=(n + 1,))
        total = int(self.off[-1])
        if total:
            buf = (ctypes.c_char * (total * 16)).from_address(ops_p.value)
            self.ops = np.frombuffer(buf, dtype=CIGAR_DTYPE, count=total)
        else:
            self.ops = np.zeros(0, dtype=CIGAR_DTYPE)

    def copy(self):
        return self.scores[:self.n].copy(), self.ops.copy(), self.off.copy()

    def free(self):
        if self._ops_p is not None:
            L = lib()
            self.ops = self.off = None
            L.gnx_free(self._ops_p)
            L.gnx_free(self._off_p)
            self._ops_p = self._off_p = None


def align_batch_windows_raw(params, a_buf, a_start, a_len, b_buf, b_start, b_len, scores=None):
    """gnx_align_batch_windows without the binding's extra copy of the results (arguments must already be contiguous uint8 / int64
    arrays).  Returns a RawResult; call .free() when done."""
    L = lib()
    n = int(a_start.shape[0])
    if scores is None:
        scores = np.zeros(max(n, 1), dtype=np.int64)
    ops_p, off_p = ctypes.c_void_p(), ctypes.c_void_p()
    check(L.gnx_align_batch_windows(ctypes.byref(params), n, a_buf.ctypes.data, a_buf.shape[0], a_start.ctypes.data, a_len.ctypes.data,
                                    b_buf.ctypes.data, b_buf.shape[0], b_start.ctypes.data, b_len.ctypes.data,
                                    scores.ctypes.data, ctypes.byref(ops_p), ctypes.byref(off_p)))
    return RawResult(scores, ops_p, off_p, n)


def init_devices(devices=None, workspace_bytes=0):
    """One context per listed device in this process (gnx_init_devices); devices=None: every visible GPU."""
    L = lib()
    if devices is None:
        check(L.gnx_init_devices(0, None, int(workspace_bytes)))
    else:
        d = np.ascontiguousarray(devices, dtype=np.int32)
        check(L.gnx_init_devices(int(d.shape[0]), d.ctypes.data, int(workspace_bytes)))
    return L.gnx_n_devices()


def set_reference(ref):
    ref = _u8(ref)
    check(lib().gnx_set_reference(ref.ctypes.data, ref.shape[0]))


def reference_info():
    """(bases, device bytes per context, 64-base blocks on the exception list) of the resident, packed reference"""
    L = lib()
    a, b, c = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    L.gnx_reference_info.argtypes = [ctypes.POINTER(ctypes.c_int64)] * 3
    L.gnx_reference_info.restype = ctypes.c_int
    check(L.gnx_reference_info(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
    return a.value, b.value, c.value


def synthetic_reference_bases(start, length, seed):
    """Host restatement of gnx_set_reference_synthetic for bases [start, start + length) (tests and benchmarks build their oracle
    inputs from it)."""
    return synthetic_reference_positions(np.arange(start, start + length, dtype=np.int64), seed)


def synthetic_reference_positions(pos, seed):
    """... for arbitrary positions"""
    pos = np.asarray(pos).astype(np.uint64)
    w = pos >> np.uint64(5)
    with np.errstate(over="ignore"):
        x = (np.uint64(seed) ^ w) + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    b = ((x >> (np.uint64(2) * (pos & np.uint64(31)))) & np.uint64(3)).astype(np.uint8)
    p = pos.astype(np.int64)
    b[(p % 50000000 < 1000) & (p >= 50000000)] = 4
    return b


def align_batch_by_offset(params, a_cat, a_off, ref_start, ref_len):
    """Reads (concatenated, a_off[n+1]) against windows of the resident reference.  Returns (scores, ops, off)."""
    L = lib()
    a_cat, a_off, ref_start, ref_len = _u8(a_cat), _i64(a_off), _i64(ref_start), _i64(ref_len)
    n = int(ref_start.shape[0])
    scores = np.zeros(max(n, 1), dtype=np.int64)
    ops_p, off_p = ctypes.c_void_p(), ctypes.c_void_p()
    check(L.gnx_align_batch_by_offset(ctypes.byref(params), n, a_cat.ctypes.data, a_off.ctypes.data, ref_start.ctypes.data, ref_len.ctypes.data,
                                      scores.ctypes.data, ctypes.byref(ops_p), ctypes.byref(off_p)))
    ops, off = _take(ops_p, off_p, n)
    return scores[:n], ops, off


def align_batch(params, alphas, betas):
    """Batch of independent pairs given as lists of uint8 arrays (concatenated form of the C ABI)."""
    L = lib()
    n = len(alphas)
    a_off = np.zeros(n + 1, dtype=np.int64)
    b_off = np.zeros(n + 1, dtype=np.int64)
    if n:
        a_off[1:] = np.cumsum([len(a) for a in alphas])
        b_off[1:] = np.cumsum([len(b) for b in betas])
    a_cat = _u8(np.concatenate([_u8(a) for a in alphas])) if n and a_off[-1] else np.zeros(1, dtype=np.uint8)
    b_cat = _u8(np.concatenate([_u8(b) for b in betas])) if n and b_off[-1] else np.zeros(1, dtype=np.uint8)
    scores = np.zeros(max(n, 1), dtype=np.int64)
    ops_p, off_p = ctypes.c_void_p(), ctypes.c_void_p()
    check(L.gnx_align_batch(ctypes.byref(params), n, a_cat.ctypes.data, a_off.ctypes.data, b_cat.ctypes.data, b_off.ctypes.data,
                            scores.ctypes.data, ctypes.byref(ops_p), ctypes.byref(off_p)))
    ops, off = _take(ops_p, off_p, n)
    return scores[:n], ops, off


def affine_gap_chunk_batch(params, chunk_size, alphas, betas):
    """align.AffineGapChunk over a batch of pairs.  Returns (scores, ops, off)."""
    L = lib()
    n = len(alphas)
    a_off = np.zeros(n + 1, dtype=np.int64)
    b_off = np.zeros(n + 1, dtype=np.int64)
    if n:
        a_off[1:] = np.cumsum([len(a) for a in alphas])
        b_off[1:] = np.cumsum([len(b) for b in betas])
    a_cat = _u8(np.concatenate([_u8(a) for a in alphas] + [np.zeros(1, np.uint8)]))
    b_cat = _u8(np.concatenate([_u8(b) for b in betas] + [np.zeros(1, np.uint8)]))
    scores = np.zeros(max(n, 1), dtype=np.int64)
    ops_p, off_p = ctypes.c_void_p(), ctypes.c_void_p()
    check(L.gnx_affine_gap_chunk_batch(ctypes.byref(params), int(chunk_size), n, a_cat.ctypes.data, a_off.ctypes.data, b_cat.ctypes.data,
                                       b_off.ctypes.data, scores.ctypes.data, ctypes.byref(ops_p), ctypes.byref(off_p)))
    ops, off = _take(ops_p, off_p, n)
    return scores[:n], ops, off


def multiple_affine_gap_batch(params, chunk_size, groups, pairs):
    """groups: list of 2-D uint8 arrays (nseq x len, alignment blocks); pairs: list of (a, b) group indices.
    align.multipleAffineGap (chunk_size 1) / multipleAffineGapChunk for every pair.  Returns (scores, ops, off)."""
    L = lib()
    g = len(groups)
    blocks = [np.ascontiguousarray(x, dtype=np.uint8).reshape(x.shape[0], -1) for x in groups]
    g_off = np.zeros(g + 1, dtype=np.int64)
    if g:
        g_off[1:] = np.cumsum([b.size for b in blocks])
    g_nseq = np.asarray([b.shape[0] for b in blocks] + [0], dtype=np.int32)
    g_len = np.asarray([b.shape[1] for b in blocks] + [0], dtype=np.int64)
    bases = _u8(np.concatenate([b.reshape(-1) for b in blocks] + [np.zeros(1, np.uint8)]))
    n = len(pairs)
    pa = np.asarray([a for a, _ in pairs] + [0], dtype=np.int32)
    pb = np.asarray([b for _, b in pairs] + [0], dtype=np.int32)
    scores = np.zeros(max(n, 1), dtype=np.int64)
    ops_p, off_p = ctypes.c_void_p(), ctypes.c_void_p()
    check(L.gnx_multiple_affine_gap_batch(ctypes.byref(params), int(chunk_size), g, bases.ctypes.data, g_off.ctypes.data, g_nseq.ctypes.data,
                                          g_len.ctypes.data, n, pa.ctypes.data, pb.ctypes.data, scores.ctypes.data,
                                          ctypes.byref(ops_p), ctypes.byref(off_p)))
    ops, off = _take(ops_p, off_p, n)
    return scores[:n], ops, off


GNX_GSW_LEFT, GNX_GSW_RIGHT = 0, 1


def gsw_extend_batch(side, scores, gap_pen, alphas, betas):
    """genomeGraph.LeftDynamicAln / RightDynamicAln (side GNX_GSW_LEFT / GNX_GSW_RIGHT) over a batch of (target, read) pairs.
    Returns (scores, end_i, end_j, ops, off); ops are the runs in traceback order, op codes 0/1/2 = 'M'/'I'/'D'."""
    L = lib()
    n = len(alphas)
    sc = np.ascontiguousarray(np.asarray(scores, dtype=np.int64).reshape(25))
    a_off = np.zeros(n + 1, dtype=np.int64)
    b_off = np.zeros(n + 1, dtype=np.int64)
    if n:
        a_off[1:] = np.cumsum([len(a) for a in alphas])
        b_off[1:] = np.cumsum([len(b) for b in betas])
    a_cat = _u8(np.concatenate([_u8(a) for a in alphas] + [np.zeros(1, np.uint8)]))
    b_cat = _u8(np.concatenate([_u8(b) for b in betas] + [np.zeros(1, np.uint8)]))
    out = np.zeros((3, max(n, 1)), dtype=np.int64)
    ops_p, off_p = ctypes.c_void_p(), ctypes.c_void_p()
    check(L.gnx_gsw_extend_batch(int(side), sc.ctypes.data, int(gap_pen), n, a_cat.ctypes.data, a_off.ctypes.data, b_cat.ctypes.data,
                                 b_off.ctypes.data, out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data,
                                 ctypes.byref(ops_p), ctypes.byref(off_p)))
    ops, off = _take(ops_p, off_p, n)
    return out[0, :n], out[1, :n], out[2, :n], ops, off


def align_pair(params, alpha, beta):
    L = lib()
    a, b = _u8(alpha), _u8(beta)
    score = ctypes.c_int64()
    nops = ctypes.c_int64()
    ops_p = ctypes.c_void_p()
    a_ptr = a.ctypes.data if a.size else None
    b_ptr = b.ctypes.data if b.size else None
    check(L.gnx_align_pair(ctypes.byref(params), a_ptr, a.shape[0], b_ptr, b.shape[0], ctypes.byref(score), ctypes.byref(ops_p), ctypes.byref(nops)))
    total = nops.value
    if total:
        buf = (ctypes.c_char * (total * 16)).from_address(ops_p.value)
        ops = np.frombuffer(buf, dtype=CIGAR_DTYPE, count=total).copy()
    else:
        ops = np.zeros(0, dtype=CIGAR_DTYPE)
    L.gnx_free(ops_p)
    return score.value, ops


def get_timing():
    t = GnxTiming()
    check(lib().gnx_get_timing(ctypes.byref(t)))
    return {"fill_ms": t.fill_ms, "traceback_ms": t.traceback_ms, "total_ms": t.total_ms, "cells": t.cells,
            "n_launches": t.n_launches, "trace_bytes": t.trace_bytes, "dominant_ms": t.dominant_ms,
            "dominant_launches": t.dominant_launches, "fast_path": t.fast_path, "host_ms": t.host_ms, "stage0_ms": t.stage0_ms,
            "fetch_ms": t.fetch_ms, "transport": t.transport, "n_contexts": t.n_contexts, "gather_ms": t.gather_ms, "bcast_ms": t.bcast_ms}


def _cat(seqs):
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    if seqs:
        off[1:] = np.cumsum([len(x) for x in seqs])
    cat = _u8(np.concatenate([_u8(x) for x in seqs] + [np.zeros(1, np.uint8)]))
    return cat, off


def seed_index_build(node_seqs, seed_len, seed_step):
    """k-mers inside nodes -> (keys uint64 sorted, locs uint64) (genomeGraph.IndexGenomeIntoMap without the node-border k-mers)"""
    L = lib()
    cat, off = _cat(node_seqs)
    kp, lp, n = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int64()
    check(L.gnx_seed_index_build(cat.ctypes.data, off.ctypes.data, len(node_seqs), int(seed_len), int(seed_step), ctypes.byref(kp), ctypes.byref(lp), ctypes.byref(n)))
    k = n.value
    if k == 0:
        return np.zeros(0, np.uint64), np.zeros(0, np.uint64)
    keys = np.ctypeslib.as_array(ctypes.cast(kp, ctypes.POINTER(ctypes.c_uint64)), shape=(k,)).copy()
    locs = np.ctypeslib.as_array(ctypes.cast(lp, ctypes.POINTER(ctypes.c_uint64)), shape=(k,)).copy()
    L.gnx_free(kp)
    L.gnx_free(lp)
    return keys, locs


GIRAF_DTYPE = np.dtype([("q_start", np.int64), ("q_end", np.int64), ("t_start", np.int64), ("t_end", np.int64), ("aln_score", np.int64),
                        ("node_off", np.int64), ("n_nodes", np.int64), ("cigar_off", np.int64), ("n_cigar", np.int64),
                        ("pos_strand", np.int32), ("flag", np.int32), ("map_q", np.int32), ("has_cigar", np.int32), ("seq_is_rc", np.int32), ("panicked", np.int32)])


class GswGraph:
    """gnx_gsw_graph: nodes, edges and the seed index of a genome graph behind the C ABI (resident on the device between batches)"""

    def __init__(self, node_seqs, edges, seed_len, seed_step):
        L = lib()
        cat, off = _cat(node_seqs)
        ef = np.ascontiguousarray([u for u, _ in edges], dtype=np.int32)
        et = np.ascontiguousarray([v for _, v in edges], dtype=np.int32)
        h = ctypes.c_void_p()
        check(L.gnx_gsw_graph_create(cat.ctypes.data, off.ctypes.data, len(node_seqs), ef.ctypes.data if len(edges) else None, et.ctypes.data if len(edges) else None,
                                     len(edges), int(seed_len), int(seed_step), ctypes.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().gnx_gsw_graph_free(self._h)
            self._h = None

    __del__ = close

    def map_reads(self, read_seqs, scores, gap_pen=-600, paired=False, threads=0):
        """-> (girafs: structured array GIRAF_DTYPE, node ids uint32, cigars CIGAR_DTYPE with op = ord('M' / 'I' / 'D' / 'S'))"""
        L = lib()
        if isinstance(read_seqs, tuple):  # (bases uint8 concatenated, offsets int64[n + 1]): as the C ABI takes them
            rcat, roff = np.ascontiguousarray(read_seqs[0], dtype=np.uint8), np.ascontiguousarray(read_seqs[1], dtype=np.int64)
            if rcat.shape[0] == 0:
                rcat = np.zeros(1, np.uint8)
            n = roff.shape[0] - 1
        else:
            rcat, roff = _cat(read_seqs)
            n = len(read_seqs)
        sc = np.ascontiguousarray(scores, dtype=np.int64).reshape(25)
        gp, np_, cp = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        check(L.gnx_gsw_map_reads(self._h, rcat.ctypes.data, roff.ctypes.data, n, 1 if paired else 0, sc.ctypes.data, int(gap_pen), int(threads),
                                  ctypes.byref(gp), ctypes.byref(np_), ctypes.byref(cp)))
        _seed_set[0] = None  # (the device's resident index is this graph's now)

        def take(ptr, dtype, count):
            if count == 0:
                return np.zeros(0, dtype=dtype)
            buf = (ctypes.c_char * (count * dtype.itemsize)).from_address(ptr.value)
            return np.frombuffer(buf, dtype=dtype, count=count).copy()
        gir = take(gp, GIRAF_DTYPE, n)
        nn = int((gir["node_off"][-1] + gir["n_nodes"][-1])) if n else 0
        nc = int((gir["cigar_off"][-1] + gir["n_cigar"][-1])) if n else 0
        nodes = take(np_, np.dtype(np.uint32), nn)
        cig = take(cp, CIGAR_DTYPE, nc)
        for ptr in (gp, np_, cp):
            if ptr.value:
                L.gnx_free(ptr)
        return gir, nodes, cig


SEED_HIT_DTYPE = np.dtype([("read_start", np.int32), ("strand", np.int32), ("node", np.int32), ("node_start", np.int32), ("q_start", np.int32), ("right", np.int32)])
_seed_set = [None]


def seed_find_batch(keys, locs, node_seqs, read_seqs, seed_len):
    """per read the list of hit tuples (read_start, strand, node, node_start, q_start, right) in the reference's order of discovery"""
    L = lib()
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    locs = np.ascontiguousarray(locs, dtype=np.uint64)
    tag = (keys.ctypes.data, locs.ctypes.data, keys.shape[0], id(node_seqs), seed_len)
    if _seed_set[0] != tag:  # make the index resident once per index object
        ncat, noff = _cat(node_seqs)
        check(L.gnx_seed_index_set(keys.ctypes.data, locs.ctypes.data, keys.shape[0], ncat.ctypes.data, noff.ctypes.data, len(node_seqs), int(seed_len)))
        _seed_set[0] = tag
    rcat, roff = _cat(read_seqs)
    hp, op = ctypes.c_void_p(), ctypes.c_void_p()
    check(L.gnx_seed_find_batch(rcat.ctypes.data, roff.ctypes.data, len(read_seqs), ctypes.byref(hp), ctypes.byref(op)))
    off = np.ctypeslib.as_array(ctypes.cast(op, ctypes.POINTER(ctypes.c_int64)), shape=(len(read_seqs) + 1,)).copy()
    total = int(off[-1])
    if total:
        buf = (ctypes.c_char * (total * SEED_HIT_DTYPE.itemsize)).from_address(hp.value)
        hits = np.frombuffer(buf, dtype=SEED_HIT_DTYPE, count=total).copy()
    else:
        hits = np.zeros(0, dtype=SEED_HIT_DTYPE)
    if hp.value:
        L.gnx_free(hp)
    L.gnx_free(op)
    return [[tuple(int(v) for v in h) for h in hits[int(off[r]):int(off[r + 1])]] for r in range(len(read_seqs))]
