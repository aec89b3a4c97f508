"""siga_amd -- MI355X-native `siga overlap` hot path (FM-index overlap engine) behind a C-ABI.

The compute lives in siga_amd/lib/libsigax.so (hand-written HIP kernels for gfx950, include/sigax.h).
This package is the thin Python side used by tests and bench.py: ctypes bindings (`_lib`) and a
mirror of the reference's OverlapBuilder interface (`overlap`).  There is no CPU fallback.
"""
from . import _lib  # noqa: F401
from .overlap import FMIndexPair, OverlapBuilder, SigaxError  # noqa: F401
