"""dna.Base encoding -- the input contract of the align hot path.

Mirrors /root/reference/dna/dna.go:5-21 (Base enum), dna/convert.go:51-84 (ByteToBase),
:143-153 (StringToBases), :178-187 (BasesToString) and dna/modify.go:8-22,60 (ToUpper/AllToUpper).
Sequences are numpy uint8 arrays of Base codes.
"""
import numpy as np

A, C, G, T, N, LowerA, LowerC, LowerG, LowerT, LowerN, Gap, Dot, Nil = range(13)

_BYTE_TO_BASE = np.full(256, 255, dtype=np.uint8)
for _ch, _b in zip("ACGTNacgtn-.*", (A, C, G, T, N, LowerA, LowerC, LowerG, LowerT, LowerN, Gap, Dot, Nil)):
    _BYTE_TO_BASE[ord(_ch)] = _b
_BASE_TO_BYTE = np.frombuffer(b"ACGTNacgtn-.*", dtype=np.uint8)
_TO_UPPER = np.arange(256, dtype=np.uint8)
_TO_UPPER[LowerA:LowerN + 1] = np.arange(A, N + 1, dtype=np.uint8)


def StringToBases(s):
    """dna.StringToBases: panics (ValueError here) on characters outside AaCcGgTtNn-.*"""
    raw = np.frombuffer(s.encode("ascii"), dtype=np.uint8)
    out = _BYTE_TO_BASE[raw]
    if (out == 255).any():
        bad = chr(int(raw[np.argmax(out == 255)]))
        raise ValueError("Error: '%s' is an invalid base." % bad)
    return out.copy()


def BasesToString(bases):
    bases = np.asarray(bases, dtype=np.uint8)
    if bases.size and int(bases.max()) > Nil:
        raise IndexError("index out of range: invalid dna.Base")
    return _BASE_TO_BYTE[bases].tobytes().decode("ascii")


def ToUpper(b):
    return int(_TO_UPPER[b])


def AllToUpper(bases):
    """In-place like the Go version; also returns the array for convenience."""
    bases[:] = _TO_UPPER[bases]
    return bases
