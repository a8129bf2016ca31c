"""CPU-side checks of the drop-in boundary: the C-ABI libraries load without a GPU and export every symbol that
include/*.h declares (no compute calls here)."""
import ctypes as C
import glob
import os
import re

import pytest

import hevc_testlib as T


def declared_symbols():
    syms = set()
    for h in glob.glob(os.path.join(T.ROOT, "include", "*.h")):
        text = open(h).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        syms |= set(re.findall(r"\b(x265amd_[a-zA-Z0-9_]+)\s*\(", text))
    return sorted(syms)


@pytest.mark.parametrize("depth", [8, 10])
def test_exports(depth):
    path = T.hip_path(depth)
    assert os.path.exists(path), "build the HIP libraries first (python -c 'import __graft_entry__ as g; g.build()')"
    lib = C.CDLL(path)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    assert lib.x265amd_bit_depth() == depth
    lib.x265amd_version.restype = C.c_char_p
    assert b"gfx950" in lib.x265amd_version()


def test_header_declares_enough():
    assert len(declared_symbols()) > 60
