"""The row forms of the in-loop filter entry points (include/x265amd.h: x265amd_deblock_rows, x265amd_sao_stats_rows, x265amd_sao_apply_rows,
x265amd_extend_border_rows on the GPU; x265amd_deblock_units_rows, x265amd_sao_rdo_rows on the host): what FrameFilter::processRow / processPostRow do for one
CTU row (reference: source/encoder/framefilter.cpp:559-664).  Run band after band in row order they must give exactly what the picture-wide entry points give,
which are pinned against the reference's Deblock / SAO classes elsewhere (tests/test_deblock.py, test_sao.py, test_planes.py).  The encoder object uses them when
pictures are coded in parallel (tests/test_encoder_api.py::test_frame_parallel_rules compares whole streams)."""
import ctypes as C

import numpy as np
import pytest

import hevc_testlib as T

_ptr = lambda a: a.ctypes.data_as(C.c_void_p)


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [8, 10])
def test_deblock_rows_equal_picture(depth):
    import torch
    L = T.load_hip(depth)
    for (seed, w, h) in ((1, 200, 136), (2, 320, 264), (3, 64, 200)):
        c = T.deblock_case(depth, seed, w, h)
        want = T.deblock_run_hip(L, c)
        isz = c["planes"][0].itemsize
        d = [torch.from_numpy(p.view(np.uint8).copy()).cuda() for p in c["planes"]]
        d_units = torch.from_numpy(c["units"].view(np.uint8).copy()).cuda()
        h4 = h // 4
        for y4 in range(0, h4, 16):
            # a CTU row: vertical edges, then horizontal edges of the band (passes = 3), as the filter thread of a picture issues them
            rc = L.lib.x265amd_deblock_rows(None, C.c_void_p(d[0].data_ptr() + c["org"][0] * isz), C.c_void_p(d[1].data_ptr() + c["org"][1] * isz),
                                            C.c_void_p(d[2].data_ptr() + c["org"][1] * isz), C.c_int64(c["stride"]), C.c_int64(c["cstride"]), w, h,
                                            C.c_void_p(d_units.data_ptr()), c["beta"], c["tc"], c["cb"], c["cr"], c["bypass"], 3, y4, min(h4, y4 + 16))
            assert rc == 0
        torch.cuda.synchronize()
        got = [t.cpu().numpy().view(c["planes"][0].dtype).reshape(p.shape) for t, p in zip(d, c["planes"])]
        for k in range(3):
            assert np.array_equal(got[k], want[k]), (seed, k)


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [8, 10])
def test_sao_rows_equal_picture(depth):
    import torch
    L = T.load_hip(depth)
    for (seed, w, h) in ((1, 200, 136), (5, 136, 192), (7, 384, 264)):
        c = T.sao_case(depth, seed, w, h)
        want = T.sao_run_hip(L, c)
        isz = c["rec"][0].itemsize
        d_rec = [torch.from_numpy(p.view(np.uint8).copy()).cuda() for p in c["rec"]]
        d_fenc = [torch.from_numpy(p.view(np.uint8).copy()).cuda() for p in c["fenc"]]
        d_out = [torch.from_numpy(p.view(np.uint8).copy()).cuda() for p in c["rec"]]
        tab = lambda ds: np.array([ds[0].data_ptr() + c["org"][0] * isz, ds[1].data_ptr() + c["org"][1] * isz, ds[2].data_ptr() + c["org"][1] * isz], np.uint64)
        n = c["nctu"] * 3 * 5 * 32
        d_cnt = torch.zeros(n, dtype=torch.int32, device="cuda"); d_org = torch.zeros(n, dtype=torch.int32, device="cuda")
        d_par = torch.from_numpy(c["params"].view(np.uint8).copy()).cuda()
        rows = (h + 63) // 64
        for r in range(rows):
            assert L.lib.x265amd_sao_stats_rows(None, _ptr(tab(d_rec)), _ptr(tab(d_fenc)), C.c_int64(c["stride"]), C.c_int64(c["cstride"]), w, h,
                                                C.c_void_p(d_cnt.data_ptr()), C.c_void_p(d_org.data_ptr()), r, r + 1) == 0
            assert L.lib.x265amd_sao_apply_rows(None, _ptr(tab(d_rec)), _ptr(tab(d_out)), C.c_int64(c["stride"]), C.c_int64(c["cstride"]), w, h,
                                                C.c_void_p(d_par.data_ptr()), r, r + 1) == 0
        torch.cuda.synchronize()
        dt = c["rec"][0].dtype
        assert np.array_equal(d_cnt.cpu().numpy(), want[0]) and np.array_equal(d_org.cpu().numpy(), want[1]), seed
        for t, p, wv in zip(d_out, c["rec"], want[2]):
            assert np.array_equal(t.cpu().numpy().view(dt).reshape(p.shape), wv), seed


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [8, 10])
def test_extend_border_rows_equal_picture(depth):
    import torch
    L = T.load_hip(depth)
    for c in T.plane_cases(depth, 11):
        isz = c["buf"].itemsize
        whole = torch.from_numpy(c["buf"].view(np.uint8).copy()).cuda()
        bands = torch.from_numpy(c["buf"].view(np.uint8).copy()).cuda()
        args = (C.c_int64(c["stride"]), c["w"], c["h"], c["mx"], c["my"])
        assert L.lib.x265amd_extend_pic_border(None, C.c_void_p(whole.data_ptr() + c["org"] * isz), *args) == 0
        for y0 in range(0, c["h"], 64):
            assert L.lib.x265amd_extend_border_rows(None, C.c_void_p(bands.data_ptr() + c["org"] * isz), *args, y0, min(c["h"], y0 + 64)) == 0
        torch.cuda.synchronize()
        assert torch.equal(whole, bands)


def test_deblock_units_rows_equal_picture():
    """host code: the deblocking records of a picture, band by band (random decisions and motion)"""
    L = T.load_hip(8)          # the library loads without a GPU; these entry points are host code
    for seed in (1, 2):
        for st in (0, 1):
            c = T.cabac_case(seed, 200, 136, st)
            w4, h4 = 200 // 4, 136 // 4
            rng = np.random.default_rng(seed)
            motion = np.zeros(w4 * h4, T.MV_UNIT_DT)
            motion["pred_mode"] = c["units"]["pred_mode"].reshape(-1)
            motion["inter_dir"] = rng.integers(1, 4 if st == 0 else 2, w4 * h4)
            motion["ref_idx"] = rng.integers(0, 2, (w4 * h4, 2))
            motion["mv"] = rng.integers(-40, 41, (w4 * h4, 2, 2))
            info = np.zeros(1, T.MVPRED_INFO_DT)
            info["pic_width"], info["pic_height"], info["num_ref_idx"] = 200, 136, (2, 2 if st == 0 else 0)
            info["ref_poc"][0, 0, :2] = (4, 2); info["ref_poc"][0, 1, :2] = (8, 4)
            si = np.array([c["si"]], T.SLICE_INFO_DT)
            units = np.ascontiguousarray(c["units"].reshape(-1))
            whole = np.zeros(w4 * h4, T.DB_UNIT_DT); bands = np.zeros(w4 * h4, T.DB_UNIT_DT)
            assert L.lib.x265amd_deblock_units(_ptr(si), _ptr(info), _ptr(units), _ptr(motion), _ptr(whole)) == 0
            for y in range(0, h4, 16):
                assert L.lib.x265amd_deblock_units_rows(_ptr(si), _ptr(info), _ptr(units), _ptr(motion), _ptr(bands), y, min(h4, y + 16)) == 0
            assert whole.tobytes() == bands.tobytes()


def test_sao_rdo_rows_equal_picture():
    """host code: the SAO decision of a picture row by row (every CTU row owns its entropy state; a CTU looks at its left and upper neighbours' parameters)"""
    L = T.load_hip(8)
    L.lib.x265amd_sao_rdo.argtypes = L.lib.x265amd_sao_rdo_rows.argtypes = None
    for seed in (3, 4):
        W, H = 264, 200
        c = T.cabac_case(seed, W, H, 1)
        rng = np.random.default_rng(seed)
        nctu = ((W + 63) // 64) * ((H + 63) // 64)
        count = rng.integers(0, 400, nctu * 3 * 5 * 32).astype(np.int32)
        org = (rng.integers(-3, 4, count.shape) * count).astype(np.int32)
        si = np.array([c["si"]], T.SLICE_INFO_DT)
        units = np.ascontiguousarray(c["units"].reshape(-1))
        out = []
        for rows in (None, 1):
            params = np.zeros(nctu, T.SAO_CTU_DT); flags = np.zeros(2, np.int32); rate = np.zeros(8, np.float64)
            if rows is None:
                assert L.lib.x265amd_sao_rdo(_ptr(si), 1, 3, 0, 69, _ptr(units), _ptr(count), _ptr(org), _ptr(rate), _ptr(params), _ptr(flags)) == 0
            else:
                for r in range((H + 63) // 64):
                    assert L.lib.x265amd_sao_rdo_rows(_ptr(si), 1, 3, 0, 69, _ptr(units), _ptr(count), _ptr(org), _ptr(rate), _ptr(params), _ptr(flags), r, r + 1) == 0
            out.append((params.tobytes(), flags.tobytes()))
        assert out[0] == out[1]
        assert np.frombuffer(out[0][0], T.SAO_CTU_DT)["type"].max() >= 0, "the decision switched SAO off everywhere: the case tests nothing"
