"""GPU edge cases of the drop-in boundary: empty batches, the installed primitive table (x265amd_setup_primitives), extreme
QPs / all-zero and saturated residuals in the TU chain, neighbour-less and fully available intra blocks, PUs on the picture
border with search windows reaching into the padding."""
import ctypes as C

import numpy as np
import pytest

import hevc_testlib as T

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("depth", [8, 10])
def test_empty_batches(depth):
    L = T.load_hip(depth)
    assert L.lib.x265amd_run_jobs(None, None, 0, 0) == 0
    assert L.lib.x265amd_tu_chain(None, None, 0, None) == 0
    assert L.lib.x265amd_intra_scan(None, None, 0, None, None) == 0
    assert L.lib.x265amd_motion_compensation(None, None, C.c_int64(0), C.c_int64(0), 0, 0, None, 0) == 0
    me = T.HipME(depth)
    assert me.lib.x265amd_me_search(me.ctx, None, C.c_void_p(8), C.c_void_p(8), C.c_int64(64), C.c_void_p(8), 0, C.c_void_p(8), C.c_void_p(8), 192, 192, 0, None, C.c_int64(0)) == 0
    # bad arguments are reported, not ignored
    assert L.lib.x265amd_run_jobs(None, None, 5, 0) < 0
    L.lib.x265amd_last_error.restype = C.c_char_p
    assert b"bad arguments" in L.lib.x265amd_last_error()
    me.close()


@pytest.mark.parametrize("depth", [8, 10])
def test_installed_primitive_table(depth):
    """x265amd_setup_primitives fills a table with the reference's EncoderPrimitives layout; call a few slots through it"""
    L, orc = T.load_hip(depth), T.load_oracle(depth)
    L.lib.x265amd_primitives_table_bytes.restype = C.c_size_t
    nbytes = L.lib.x265amd_primitives_table_bytes()
    assert nbytes == 2281 * 8
    table = (C.c_void_p * 2281)()
    assert L.lib.x265amd_setup_primitives(table, C.c_size_t(nbytes - 8)) < 0        # wrong size is refused
    nset = L.lib.x265amd_setup_primitives(table, C.c_size_t(nbytes))
    assert nset > 900     # 943 slots on the north-star path
    rng = np.random.default_rng(3)
    a, b = T.pix_buf(L, rng, 64 * 64, "random"), T.pix_buf(L, rng, 80 * 64, "random")
    PU_FIELDS, CU_FIELDS, BASE_CU = 19, 73, 25 * 19
    cmp_t = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64)
    for part in (1, 4, 13, 22):                                                      # pu[part].sad (field 0), pu[part].satd (field 4)
        for field, name in ((0, "sad"), (4, "satd")):
            fn = cmp_t(table[part * PU_FIELDS + field])
            assert fn(T._ptr(a), 64, T._ptr(b), 80) == orc.call(name, part, a, 64, b, 80), (part, name)
    for cu in (1, 3):                                                                # cu[cu].sa8d: ordinal 28 (CU_sa8d in host/primitive_table.h)
        fn = cmp_t(table[BASE_CU + cu * CU_FIELDS + 28])
        assert fn(T._ptr(a), 64, T._ptr(b), 80) == orc.call("sa8d", cu, a, 64, b, 80)
    dct_t = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int64)
    src = T.s16_buf(rng, 32 * 40, -L.pmax, L.pmax, "random")
    for cu in (0, 2, 3):                                                             # cu[cu].dct: ordinal 0
        n = 4 << cu
        got, want = np.zeros(n * n, np.int16), np.zeros(n * n, np.int16)
        dct_t(table[BASE_CU + cu * CU_FIELDS + 0])(T._ptr(src), T._ptr(got), 40)
        orc.call("dct", cu, src, want, 40)
        assert np.array_equal(got, want)
    # a slot that is not on the path stays untouched (NULL in our zeroed table): pu[part].ads (field 3)
    assert table[3] is None


@pytest.mark.parametrize("depth", [8, 10])
def test_tu_chain_extremes(depth):
    """QP 0 and 51(+offset), zero residual, maximal residual, every size / text type / intra-inter / slice type"""
    import torch
    hip, orc = T.load_hip(depth), T.load_oracle(depth)
    pmax = (1 << depth) - 1
    dt = np.uint8 if depth == 8 else np.uint16
    cases = []
    for log2 in (2, 3, 4, 5):
        N = 1 << log2
        for qp in (0, 51):
            for kind in range(4):
                fenc = np.full((N, N), pmax if kind in (1, 3) else pmax // 2, dt)
                pred = {0: fenc.copy(), 1: np.zeros((N, N), dt), 2: np.full((N, N), pmax // 2 + 1, dt),
                        3: (np.indices((N, N)).sum(0) % 2 * pmax).astype(dt)}[kind]
                for intra in (0, 1):
                    cases.append(dict(fenc=fenc, pred=pred, log2=log2, ttype=kind % 3, intra=intra, dir=10 + 16 * intra, slice=2 if intra else kind % 2,
                                      qp=qp + 6 * (depth - 8), signhide=1))
    want = T.tu_run_chain_oracle(orc, cases)
    isz = np.dtype(dt).itemsize
    per = 32 * 32 * (isz * 3 + 4)
    arena = np.zeros(len(cases) * per, np.uint8)
    jobs = np.zeros(len(cases), T.TU_JOB_DT)
    for i, c in enumerate(cases):
        N = 1 << c["log2"]
        arena[i * per:i * per + 1024 * isz].view(dt).reshape(32, 32)[:N, :N] = c["fenc"]
        arena[i * per + 1024 * isz:i * per + 2048 * isz].view(dt).reshape(32, 32)[:N, :N] = c["pred"]
    d_arena = torch.from_numpy(arena).cuda()
    for i, c in enumerate(cases):
        base = d_arena.data_ptr() + i * per
        jobs[i] = (base, base + 1024 * isz, base + 3072 * isz, base + 3072 * isz + 2048, base + 2048 * isz, 32, 32, 32, 32,
                   c["log2"], c["ttype"], c["intra"], c["dir"], c["slice"], c["qp"], c["signhide"], 0)
    d_jobs = torch.from_numpy(jobs.view(np.uint8).copy()).cuda()
    d_out = torch.zeros(len(cases) * 32, dtype=torch.uint8, device="cuda")
    assert hip.lib.x265amd_tu_chain(None, C.c_void_p(d_jobs.data_ptr()), len(cases), C.c_void_p(d_out.data_ptr())) == 0
    torch.cuda.synchronize()
    res = d_out.cpu().numpy().view(T.TU_RESULT_DT)
    back = d_arena.cpu().numpy()
    for i, (c, w) in enumerate(zip(cases, want)):
        N = 1 << c["log2"]
        got = (int(res[i]["num_sig"]), int(res[i]["zero_dist"]), int(res[i]["zero_energy"]), int(res[i]["nz_dist"]), int(res[i]["nz_energy"]))
        assert got == tuple(int(v) for v in w[0]), (i, {k: v for k, v in c.items() if k not in ("fenc", "pred")})
        assert np.array_equal(back[i * per + 2048 * isz:i * per + 3072 * isz].view(dt).reshape(32, 32)[:N, :N], w[3]), i


@pytest.mark.parametrize("depth", [8, 10])
def test_me_picture_border(depth):
    """PUs in the picture corners: search windows and 8-tap margins reach into the padding, exactly as in the reference"""
    me, orc = T.HipME(depth), T.load_oracle(depth)
    cur, rp, stride, origin = T.me_make_planes(depth, 21, motion=(3, 2))
    jobs = []
    for (x, y) in ((0, 0), (192, 0), (0, 128), (192, 128), (248, 184), (0, 184)):
        for (w, h) in ((64, 64), (8, 8), (16, 8)):
            if x + w > 256 or y + h > 192 or (x % 64) + w > 64 or (y % 64) + h > 64:
                continue
            for method in (T.ME_HEX, T.ME_STAR, T.ME_DIA):
                mnx, mxx = max(-57, -x - 64), min(57, 256 - x - w + 64)
                mny, mxy = max(-57, -y - 56), min(57, 192 - y - h + 56)
                jobs.append(dict(x=x, y=y, w=w, h=h, qp=30, mvp=(-40, 37), mvmin=(mnx, mny), mvmax=(mxx, mxy), mvc=[(200, -180), (-220, 210)],
                                 merange=57, method=method, subme=3))
    want = T.me_run_host(orc, cur, rp, stride, origin, jobs)
    got = me.run(cur, rp, stride, origin, jobs)
    assert np.array_equal(want, got)
    me.close()
