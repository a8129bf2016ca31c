"""x265-amod_amd: MI355X-native HEVC encode hot path behind the x265 primitive-table boundary.

The product is the C-ABI shared library built from csrc/ (include/x265amd.h); this Python package is only the
thin loader used by tests, bench.py and __graft_entry__.py.  It never falls back to a CPU implementation: loading
fails loudly if the HIP library has not been built.
"""
from .capi import load, lib_path, X265AmdError  # noqa: F401
from . import gop_shard  # noqa: F401
from . import frame_rows  # noqa: F401
