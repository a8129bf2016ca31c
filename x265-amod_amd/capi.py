"""ctypes loader for libx265amd_main.so / libx265amd_main10.so (see include/x265amd.h)."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))


class X265AmdError(RuntimeError):
    pass


def lib_path(depth=8):
    return os.path.join(HERE, "lib", "libx265amd_main.so" if depth == 8 else "libx265amd_main10.so")


class Job(C.Structure):
    """mirror of struct x265amd_job (include/x265amd.h)"""
    _fields_ = [("op", C.c_int32), ("size", C.c_int32), ("p", C.c_int32 * 6),
                ("a", C.c_uint64), ("b", C.c_uint64), ("c", C.c_uint64), ("d", C.c_uint64),
                ("e", C.c_uint64 * 2), ("sa", C.c_int32), ("sb", C.c_int32), ("sc", C.c_int32), ("sd", C.c_int32)]


_libs = {}


def load(depth=8):
    """Returns the ctypes handle of the HIP library for `depth`; raises if it is missing (no fallback)."""
    if depth in _libs:
        return _libs[depth]
    path = lib_path(depth)
    if not os.path.exists(path):
        raise X265AmdError("HIP library %s not built: run x265-amod_amd/build.sh (there is no CPU fallback)" % path)
    try:
        import torch  # noqa: F401  (load torch's HIP runtime first; see tests/hevc_testlib.py::load_hip)
    except ImportError:
        pass
    lib = C.CDLL(path)
    lib.x265amd_version.restype = C.c_char_p
    lib.x265amd_last_error.restype = C.c_char_p
    if lib.x265amd_bit_depth() != depth:
        raise X265AmdError("bit depth mismatch in %s" % path)
    _libs[depth] = lib
    return lib


def check(lib, rc, what):
    if rc < 0:
        raise X265AmdError("%s failed (%d): %s" % (what, rc, lib.x265amd_last_error().decode()))
    return rc
