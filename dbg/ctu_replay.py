"""Runs the reference's Analysis::compressCTU (oracle/_ref) on the states dumped by X265AMD_DUMP_CTU and compares with what the product produced."""
import sys, os, glob
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, ctypes as C, hevc_testlib as T
for path in sorted(glob.glob(os.path.join(sys.argv[1], "ctu_*.bin"))):
    b = np.fromfile(path, np.uint8)
    o = 0
    def take(dt, n=1):
        global o
        a = np.frombuffer(b, dt, n, o).copy(); o += a.nbytes; return a
    hdr = take(np.int32, 12)
    W, H, npics, stride, cstride, mx, my, addr, isz, nctu = [int(v) for v in hdr[:10]]
    w4, h4 = W // 4, H // 4
    info = take(T.MVPRED_INFO_DT); sp = take(T.INTER_SP_DT); si = take(T.SLICE_INFO_DT); ap = take(T.ANALYSIS_PARAMS_DT)
    units = take(T.CU_UNIT_DT, w4 * h4); cur = take(T.MV_UNIT_DT, w4 * h4); col = take(T.MV_UNIT_DT, w4 * h4)
    ref_depth = take(np.uint8, 2 * w4 * h4); ref_qp0 = take(np.int8, 2 * nctu); stat = take(T.CU_STAT_DT, nctu + 1)
    ctx = take(np.uint8, 160); frac = int(take(np.uint64)[0])
    dt = np.uint8 if isz == 1 else np.uint16
    planes, addrs = [], []
    for k in range(npics * 3):
        luma = k % 3 == 0
        pmx, pmy, ph, st = (mx, my, H, stride) if luma else (mx // 2, my // 2, H // 2, cstride)
        a = take(dt, (ph + 2 * pmy) * st)
        planes.append(a); addrs.append(a.ctypes.data + (pmy * st + pmx) * isz)
    res_p = take(T.CTU_RESULT_DT); units_p = take(T.CU_UNIT_DT, w4 * h4); cur_p = take(T.MV_UNIT_DT, w4 * h4); coeff_p = take(np.int16, T.RD_TILE)
    R = T.load_ref(8 if isz == 1 else 10)
    rsp = np.zeros(1, T.SEARCH_PARAMS_DT)
    rsp["searchMethod"], rsp["subpelRefine"], rsp["searchRange"], rsp["qp"], rsp["bChromaMC"], rsp["numPics"] = sp["search_method"], sp["subpel_refine"], sp["search_range"], sp["qp"], sp["chroma_mc"], npics
    rsp["refPic"] = sp["ref_pic"]
    u = units.copy(); m = cur.copy(); st_ = stat.copy()
    ctuW = (W + 63) // 64
    uo = np.zeros((16, 16), T.CU_UNIT_DT); mo = np.zeros((16, 16), T.MV_UNIT_DT); coeff = np.zeros(T.RD_TILE, np.int16); res = np.zeros(1, T.CTU_RESULT_DT)
    pl = np.array(addrs, np.uint64)
    R.lib.ref_compress_ctu(T._ptr(info), T._ptr(rsp), T._ptr(si), T._ptr(ap), T._ptr(u), T._ptr(m), T._ptr(col), T._ptr(ref_depth), T._ptr(ref_qp0), T._ptr(pl), C.c_int64(stride), C.c_int64(cstride),
                           mx, my, T._ptr(st_), addr, T._ptr(ctx), C.c_uint64(frac), T._ptr(uo), T._ptr(mo), T._ptr(coeff), T._ptr(res))
    cx, cy = (addr % ctuW) * 16, (addr // ctuW) * 16
    up = units_p.reshape(h4, w4)[cy:cy + 16, cx:cx + 16]
    hh, ww = up.shape
    same_cost = int(res["rd_cost"][0]) == int(res_p["rd_cost"][0])
    diffs = []
    for f in ("depth", "pred_mode", "part_size", "tu_depth", "cbf", "luma_dir"):
        d = np.argwhere(uo[f][:hh, :ww].reshape(hh, ww, -1) != up[f].reshape(hh, ww, -1))
        if len(d): diffs.append((f, d[:3].tolist()))
    print(os.path.basename(path), "addr", addr, "rd_cost ref/prod", int(res["rd_cost"][0]), int(res_p["rd_cost"][0]), "bits", int(res["total_bits"][0]), int(res_p["total_bits"][0]),
          "coeff equal", bool(np.array_equal(coeff, coeff_p)), "diffs", diffs)
    if diffs:
        y, x = diffs[0][1][0][:2]
        print("   at unit (y,x)", y, x, "ref", [int(uo[f][y, x]) if uo[f][y, x].ndim == 0 else uo[f][y, x].tolist() for f in ("depth", "pred_mode", "part_size", "tu_depth", "cbf", "luma_dir", "merge_flag", "inter_dir")],
              "prod", [int(up[f][y, x]) if up[f][y, x].ndim == 0 else up[f][y, x].tolist() for f in ("depth", "pred_mode", "part_size", "tu_depth", "cbf", "luma_dir", "merge_flag", "inter_dir")])
