"""gandalf_amd: MI355X-native (gfx950) GANDALF grad-h SPH + KD-tree gravity hot path.

The product is the C-ABI shared library `gandalf_amd/csrc/libgandalf_hip.so` (declared in
include/gandalf_hip.h).  This package is the thin ctypes binding the tests and bench.py use; it has
no CPU fallback - importing the binding without the built library raises.
"""
from .capi import GandalfHip, Config, Stats, load_library, LIB_PATH, GhError, FIELDS  # noqa: F401
