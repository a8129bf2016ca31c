"""x265amd_intra_cu_bits (include/x265amd.h, csrc/intra_cu_dev.h): the bits of a decided intra CU counted by one wavefront.  Checked against the reference's own
Search::checkIntra results (tests/golden/check_intra_golden.npz, made by oracle/_ref from the fixtures of tests/test_intra_rd.py): the golden's chosen directions,
coded block flags and levels go in, and the golden's total / prediction-info / coefficient bits, the coder's fraction and the contexts left behind must come out
(search.cpp:1236-1287: skip flag + pred mode in P slices, codePartSize, codePredInfo, codeCoeff).  The fixtures with delta-QP coding switched on are left out: the
entry point has no delta QP (the encoder object runs constant QP), and the cases with transform splits inside the CU likewise."""
import ctypes as C
import os

import numpy as np
import pytest

import hevc_testlib as T
from test_intra_rd import CHECK_CASES, CHECK_GOLD_PATH

JOB_DT = np.dtype([("ctx", "u1", 160), ("frac_bits", "<u8"), ("log2_cu", "u1"), ("nxn", "u1"), ("code_part_size", "u1"), ("inter_slice", "u1"), ("skip_ctx", "u1"),
                   ("sign_hide", "u1"), ("chroma_dir", "u1"), ("cbf_u", "u1"), ("cbf_v", "u1"), ("subdiv_flag", "u1"), ("reserved", "u1", 2), ("luma_dir", "u1", 4), ("cbf_y", "u1", 4),
                   ("preds", "u1", (4, 3)), ("lev_y", "<u8", 4), ("lev_u", "<u8"), ("lev_v", "<u8")])
OUT_DT = np.dtype([("ctx", "u1", 160), ("frac_bits", "<u8"), ("mv_frac", "<u8"), ("skip_frac", "<u8")])


def test_intra_cu_bits_layout():
    assert JOB_DT.itemsize == 248 and JOB_DT.fields["lev_y"][1] == 200 and OUT_DT.itemsize == 184


def neighbour_dir(units, x4, y4, above):
    """the neighbour's luma direction as getIntraDirLumaPredictor sees it (cudata.cpp:910-953): DC when it is outside the picture, not intra, or above the CTU"""
    if above:
        if y4 % 16 == 0:
            return 1
        u = units[y4 - 1, x4]
    else:
        if x4 == 0:
            return 1
        u = units[y4, x4 - 1]
    return int(u["luma_dir"]) if u["pred_mode"] == T.MODE_INTRA else 1


def jobs_from_golden(gold, k, c, lev_addr):
    """-> (jobs, wanted results) of case k; lev_addr(i) = device address of CU i's levels (luma S*S, then U, V)"""
    n = len(c["cus"])
    jobs = np.zeros(n, JOB_DT)
    want = []
    units = c["units"]
    for i in range(n):
        cu = c["cus"][i]
        log2 = int(cu["log2_size"]); S = 1 << log2
        x4, y4 = int(cu["x"]) // 4, int(cu["y"]) // 4
        dirs = gold["%d/%d/dirs" % (k, i)]; un = gold["%d/%d/units" % (k, i)]; res = gold["%d/%d/res" % (k, i)]
        nxn = int(dirs[0, 2] == 3)
        j = jobs[i]
        j["ctx"] = cu["ctx"]; j["frac_bits"] = cu["frac_bits"]
        j["log2_cu"], j["nxn"], j["code_part_size"] = log2, nxn, int(log2 == 3)
        j["inter_slice"] = int(c["si"]["slice_type"] != 2)
        if j["inter_slice"]:
            left = x4 > 0 and units[y4, x4 - 1]["pred_mode"] == T.MODE_SKIP
            above = y4 > 0 and units[y4 - 1, x4]["pred_mode"] == T.MODE_SKIP
            j["skip_ctx"] = int(left) + int(above)
        j["sign_hide"] = int(c["si"]["sign_hide"])
        j["chroma_dir"] = int(dirs[0, 1])
        j["cbf_u"], j["cbf_v"] = int(un[0, 2]) & 1, int(un[0, 3]) & 1
        range0 = min(max(log2 - (int(c["si"]["tu_max_depth_intra"]) - 1 + nxn), int(c["si"]["tu_log2_min"])), int(c["si"]["tu_log2_max"]))     # getIntraTUQtDepthRange (cudata.cpp)
        j["subdiv_flag"] = int(not nxn and log2 > range0)
        base = lev_addr(i)
        for p in range(4 if nxn else 1):
            px, py = p & 1, p >> 1
            j["luma_dir"][p] = int(dirs[p, 0])
            j["cbf_y"][p] = (int(un[p, 1]) >> nxn) & 1
            left = int(dirs[p - 1, 0]) if (nxn and px) else neighbour_dir(units, x4, y4 + py, False)
            above = int(dirs[p - 2, 0]) if (nxn and py) else neighbour_dir(units, x4 + px, y4, True)
            j["preds"][p] = T.luma_mpm(left, above)
            j["lev_y"][p] = base + 2 * 16 * p
        j["lev_u"] = base + 2 * S * S
        j["lev_v"] = base + 2 * (S * S + S * S // 4)
        want.append(dict(split=int(un[:, 0].max()) > nxn, total=int(res[0]), mv=int(res[1]), coeff=int(res[2]), frac=int(res[9]), ctx=gold["%d/%d/ctx" % (k, i)]))
    return jobs, want


def usable(cfg):
    return len(cfg) == 5 and not int(T.check_intra_case(cfg[0], cfg[1], cfg[2], cfg[3], strong=cfg[4])["si"]["use_dqp"])


def model(O, j, lev, en_bits, lps_next):
    """the same walk on the CPU from the oracle's pieces (the estimator's tables, the oracle's bits-only coefficient coding): -> (frac, mv_frac, skip_frac, ctx)"""
    ctx = np.array(j["ctx"][:T.CTX_COUNT], np.uint8)
    frac = int(j["frac_bits"]) & 32767
    skipf = 0

    def bin_(ci, b):
        nonlocal frac
        frac += en_bits[int(ctx[ci]) ^ b]
        ctx[ci] = T.cabac_next_state(int(ctx[ci]), b, lps_next)

    def coeffs(levels, log2, ttype, dir_):
        nonlocal frac, ctx
        cc = dict(ctx=ctx, log2=log2, ttype=ttype, intra=1, dir=dir_, signhide=int(j["sign_hide"]))
        b, new = T.coeff_bits_run(O, [cc], [(1, np.ascontiguousarray(levels, np.int16))])[0]
        frac += b; ctx = np.array(new, np.uint8)

    nxn, log2 = int(j["nxn"]), int(j["log2_cu"])
    S = 1 << log2
    if j["inter_slice"]:
        bin_(3 + int(j["skip_ctx"]), 0); skipf = frac
        bin_(12, 1)
    if j["code_part_size"]:
        bin_(8, 0 if nxn else 1)
    npu = 4 if nxn else 1
    pidx = []
    for p in range(npu):
        preds = [int(x) for x in j["preds"][p]]
        d = int(j["luma_dir"][p])
        pidx.append(preds.index(d) if d in preds else -1)
        bin_(13, int(pidx[-1] != -1))
    for p in range(npu):
        frac += (1 + (pidx[p] != 0) if pidx[p] != -1 else 5) << 15
    if j["chroma_dir"] == 36:
        bin_(14, 0)
    else:
        bin_(14, 1); frac += 2 << 15
    mvf = frac
    cdir = int(j["luma_dir"][0]) if j["chroma_dir"] == 36 else int(j["chroma_dir"])
    if j["subdiv_flag"]:
        bin_(35 + 5 - log2, 0)
    bin_(30, int(j["cbf_u"])); bin_(30, int(j["cbf_v"]))
    if not nxn:
        bin_(29, int(j["cbf_y"][0]))
        if j["cbf_y"][0]:
            coeffs(lev[:S * S], log2, 0, int(j["luma_dir"][0]))
        cs = S * S // 4
    else:
        for p in range(4):
            bin_(28, int(j["cbf_y"][p]))
            if j["cbf_y"][p]:
                coeffs(lev[16 * p:16 * p + 16], 2, 0, int(j["luma_dir"][p]))
        cs = 16
    clog2 = 2 if nxn else log2 - 1
    if j["cbf_u"]:
        coeffs(lev[S * S:S * S + cs], clog2, 1, cdir)
    if j["cbf_v"]:
        coeffs(lev[S * S + S * S // 4:S * S + S * S // 4 + cs], clog2, 2, cdir)
    return frac, mvf, skipf, ctx


def test_cpu_model_of_the_cu_bits_matches_reference_golden():
    """the walk this entry point makes, from the oracle's pieces, against the reference's checkIntra results: pins the order of the bins and the contexts"""
    gold = np.load(CHECK_GOLD_PATH)
    en_bits, lps_next = T.entropy_bit_tables()
    checked = 0
    for k, cfg in enumerate(CHECK_CASES):
        if not usable(cfg):
            continue
        depth, seed, st, psy, strong = cfg[:5]
        O = T.load_oracle(depth)
        c = T.check_intra_case(depth, seed, st, psy, strong=strong)
        jobs, want = jobs_from_golden(gold, k, c, lambda i: 0)
        for i in range(len(jobs)):
            lev = np.zeros(1024 + 512, np.int16)
            co = gold["%d/%d/coeff" % (k, i)]
            lev[:co.size] = co
            w = want[i]
            if w["split"]:          # the reference split the CU's transform tree: not this entry point's form
                continue
            frac, mvf, skipf, ctx = model(O, jobs[i], lev, en_bits, lps_next)
            assert (frac, frac >> 15, (mvf >> 15) - (skipf >> 15)) == (w["frac"], w["total"], w["mv"]), (k, i, frac, mvf, skipf, w)
            assert np.array_equal(ctx, w["ctx"]), (k, i)
            checked += 1
    assert checked == 45


def test_intra_cu_bits_golden_has_the_cases():
    gold = np.load(CHECK_GOLD_PATH)
    kinds = set()
    for k, cfg in enumerate(CHECK_CASES):
        if not usable(cfg):
            continue
        for i in range(10):
            kinds.add((int(gold["%d/%d/dirs" % (k, i)][0, 2] == 3), cfg[2] != 2, bool(gold["%d/%d/coeff" % (k, i)].size)))
    assert {(0, False, True), (1, False, True), (0, True, True), (1, True, True)} <= kinds, kinds


@pytest.mark.gpu
def test_hip_intra_cu_bits_matches_reference_golden():
    import torch
    gold = np.load(CHECK_GOLD_PATH)
    checked = 0
    for k, cfg in enumerate(CHECK_CASES):
        if not usable(cfg):
            continue
        depth, seed, st, psy, strong = cfg[:5]
        H = T.load_hip(depth)
        c = T.check_intra_case(depth, seed, st, psy, strong=strong)
        n = len(c["cus"])
        lev = np.zeros((n, 1024 + 512), np.int16)
        for i in range(n):
            co = gold["%d/%d/coeff" % (k, i)]
            lev[i, :co.size] = co
        d_lev = torch.from_numpy(lev).cuda()
        jobs, want = jobs_from_golden(gold, k, c, lambda i: d_lev.data_ptr() + i * lev.shape[1] * 2)
        d_jobs = torch.from_numpy(jobs.view(np.uint8).reshape(-1).copy()).cuda()
        d_out = torch.zeros(n * OUT_DT.itemsize, dtype=torch.uint8, device="cuda")
        with T.call_stream(H) as st_:
            rc = H.lib.x265amd_intra_cu_bits(st_, C.c_uint64(d_jobs.data_ptr()), n, C.c_uint64(d_out.data_ptr()))
        assert rc == 0, H.lib.x265amd_last_error()
        out = d_out.cpu().numpy().view(OUT_DT)
        for i in range(n):
            o, w = out[i], want[i]
            if w["split"]:
                continue
            skip_bits = int(o["skip_frac"]) >> 15
            got = dict(total=int(o["frac_bits"]) >> 15, mv=(int(o["mv_frac"]) >> 15) - skip_bits, frac=int(o["frac_bits"]))
            got["coeff"] = got["total"] - got["mv"] - skip_bits
            what = "case %d CU %d (log2 %d nxn %d slice %d)" % (k, i, jobs[i]["log2_cu"], jobs[i]["nxn"], st)
            assert all(got[f] == w[f] for f in ("total", "mv", "coeff", "frac")), "%s: got %s, the reference counted %s; job %s" % (
                what, got, {f: w[f] for f in got}, {f: jobs[i][f].tolist() for f in JOB_DT.names if f not in ("ctx", "reserved")})
            assert np.array_equal(o["ctx"][:T.CTX_COUNT], w["ctx"]), "%s: contexts differ at %s" % (what, np.flatnonzero(o["ctx"][:T.CTX_COUNT] != w["ctx"])[:8])
            checked += 1
    assert checked == 45
