"""Device job queues (x265-amod_amd/csrc/xa_queue.h, device_queue.hip): the CTU rows of the encoder object run their block operations as commands to
resident workgroups instead of kernel launches.  The transport must not change a byte: the end-to-end streams of tests/test_encoder_api.py run through
the queues by default; here the transport itself is exercised, and one configuration is repeated with the queues switched off (X265AMD_QUEUES=0, in a
child process because the switch is read once) so that both paths stay pinned to the reference encoder's bytes."""
import ctypes as C
import os
import subprocess
import sys

import pytest

import hevc_testlib as T


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [8, 10])
def test_queue_selftest(depth):
    L = T.load_hip(depth)
    # copies through the BAR ring, rectangle copies, fills, deferred copies to pageable memory: 24 queues hammered by 24 host threads
    L.lib.x265amd_last_error.restype = C.c_char_p
    assert L.lib.x265amd_queue_selftest(400, 24) == 0, L.lib.x265amd_last_error().decode()
    # again: the server generation ended with the last release and starts anew
    assert L.lib.x265amd_queue_selftest(50, 3) == 0, L.lib.x265amd_last_error().decode()


@pytest.mark.gpu
def test_a_wait_that_gives_up_poisons_its_owner_only():
    """XA_OP_WAIT bounds its wait (two seconds); a queue whose wait gave up runs nothing more and fails every host wait -- for the owner it happened to: released and handed out
    again (the resident kernel still running) the queue starts clean (XA_CMD_RESET, csrc/device_queue.hip: xa_queue_clear_fault)"""
    L = T.load_hip(8)
    L.lib.x265amd_last_error.restype = C.c_char_p
    assert L.lib.x265amd_queue_selftest_wait_fault() == 0, L.lib.x265amd_last_error().decode()
    assert L.lib.x265amd_queue_selftest(20, 4) == 0, L.lib.x265amd_last_error().decode()


@pytest.mark.gpu
def test_stream_path_still_reproduces_reference_stream():
    env = dict(os.environ, X265AMD_QUEUES="0")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.join(T.ROOT, "tests", "test_encoder_api.py"),
                        "-k", "sao_bframes or wpp/ or hbd_wpp"], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_selftest_symbol_is_exported():
    assert hasattr(C.CDLL(T.hip_path(8)), "x265amd_queue_selftest")


@pytest.mark.gpu
def test_repeated_encodes_are_identical():
    """the same clip eight times in one process (warm buffer pools, the job server restarted between pictures): every stream must be the reference
    encoder's.  A transport fault shows here as an occasional different stream -- a command slot accepted half written did exactly that once."""
    import hashlib
    import numpy as np
    import test_encoder_api as E
    g = np.load(E.EDGE_GOLD)
    tag = "wvga/"
    (w, h), n, cfg = E.EDGE_CONFIGS[tag]
    want = hashlib.md5(g[tag + "stream"].tobytes()).hexdigest()
    clip = T.encoder_api_clip(tag, w, h, n, 8)
    L = T.load_hip(8)
    for r in range(8):
        stream, _ = T.encoder_run(L, clip, w, h, **cfg)
        assert hashlib.md5(stream.tobytes()).hexdigest() == want, "run %d" % r
