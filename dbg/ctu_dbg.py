import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, hevc_testlib as T, test_ctu_analysis as tca
gold = np.load(tca.GOLD_PATH)
mes = {}
for k in [int(a) for a in sys.argv[1:]]:
    cfg = (tca.CASES + tca.PART_CASES + tca.RDOQ_CASES)[k]
    depth = cfg[0]
    if depth not in mes: mes[depth] = T.HipME(depth)
    c = tca.make_case(k)
    got = T.ctu_pack(T.ctu_run_hip(T.load_hip(depth), mes[depth], c))
    for i, d in enumerate(got):
        for name, a in d.items():
            want = gold["%d/%d/%s" % (k, i, name)]
            if not np.array_equal(a, want):
                bad = np.argwhere(np.asarray(a) != want)
                print("case", k, cfg, "ctu", i, name, "nbad", len(bad), "first", bad[:4].tolist())
                if name == "units":
                    r = bad[0][0]
                    print("  unit", r, "(y,x)=", divmod(int(r), 16), "got", a[r].tolist(), "want", want[r].tolist())
    print("case", k, "done")
