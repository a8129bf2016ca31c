"""The reference calls the primitive table from several threads at once (frame encoder, lookahead, pre-lookahead workers:
threadpool.cpp / slicetype.cpp); the layer-1 shims must therefore be re-entrant.  Several host threads hammer different
slots through libx265amd concurrently; every result must equal the oracle's."""
import threading

import numpy as np
import pytest

import hevc_testlib as T

pytestmark = pytest.mark.gpu


def work_items(L, O, rng):
    """(name, callable(lib) -> comparable bytes) pairs on fixed inputs"""
    items = []
    for part in (0, 1, 4, 12):          # 8x8 .. larger PUs
        w, h = T.PU_SIZES[part]
        a = rng.integers(0, L.pmax + 1, (h, 80)).astype(L.pixel)
        b = rng.integers(0, L.pmax + 1, (h, 96)).astype(L.pixel)
        for nm in ("sad", "satd"):
            if (w % 4 == 0 and h % 4 == 0) or nm == "sad":
                items.append((nm + str(part), lambda lib, nm=nm, part=part, a=a, b=b: int(lib.call(nm, part, a, 80, b, 96))))
    for cu in range(4):
        n = 4 << cu
        nb = rng.integers(0, L.pmax + 1, 4 * n + 1 + 16).astype(L.pixel)
        for mode in (0, 1, 5, 10, 20, 26, 34):
            def f(lib, cu=cu, n=n, nb=nb, mode=mode):
                d = np.zeros(n * n, L.pixel)
                lib.call("intra_pred", cu, mode, d, n, nb, int(n <= 16))
                return d.tobytes()
            items.append(("intra%d_%d" % (cu, mode), f))
        src = rng.integers(-255, 256, (n, n)).astype(np.int16)
        def g(lib, cu=cu, n=n, src=src):
            d = np.zeros(n * n, np.int16)
            lib.call("dct", cu, src, d, n)
            return d.tobytes()
        items.append(("dct%d" % cu, g))
    for cu in (1, 2):
        n = 4 << cu
        a = rng.integers(0, L.pmax + 1, (n, 64)).astype(L.pixel)
        items.append(("var%d" % cu, lambda lib, cu=cu, a=a: int(lib.call("var", cu, a, 64))))
    return items


@pytest.mark.parametrize("depth", [8, 10])
def test_shims_from_concurrent_threads(depth):
    L, O = T.load_hip(depth), T.load_oracle(depth)
    rng = np.random.default_rng(77)
    items = work_items(L, O, rng)
    want = {nm: f(O) for nm, f in items}
    bad = []

    def worker(seed):
        r = np.random.default_rng(seed)
        for _ in range(400):
            nm, f = items[int(r.integers(0, len(items)))]
            if f(L) != want[nm]:
                bad.append(nm)

    threads = [threading.Thread(target=worker, args=(s,)) for s in range(4)]
    for t in threads: t.start()
    for t in threads: t.join()
    assert not bad, sorted(set(bad))
