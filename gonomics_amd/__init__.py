"""gonomics_amd -- MI355X (gfx950) drop-in for the pairwise-DP hot path of gonomics' `align` package.

Only what the path needs lives here:
  csrc/        hand-written HIP kernels + the C ABI (include/gnx_align.h) -> libgonomics_align_hip.so
  align.py     host-side mirror of the reference's `align` API (AffineGap, ConstGap, ... same names)
  dna.py       dna.Base encoding helpers (input contract of the path)
  fasta.py     minimal FASTA reader for the callers' fixtures
  _lib.py      ctypes binding of the C ABI (fails loudly when the HIP library is missing)
"""
from . import dna  # noqa: F401
