# the time of x265amd_sao_stats_rows_cols for a unit of 2 CTUs (what the filter thread of a picture launches per sweep), alone on the device and beside the resident
# job server (a held queue keeps it resident): dbg/sao_unit_time.py
import sys, os, time, ctypes as C
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, hevc_testlib as T
L = T.load_hip(8)
w, h = 1920, 1080
c = T.sao_case(8, 3, w, h)
isz = 1
d_rec = [torch.from_numpy(p.view(np.uint8).copy()).cuda() for p in c["rec"]]
d_fenc = [torch.from_numpy(p.view(np.uint8).copy()).cuda() for p in c["fenc"]]
tab = lambda ds: np.array([ds[0].data_ptr() + c["org"][0] * isz, ds[1].data_ptr() + c["org"][1] * isz, ds[2].data_ptr() + c["org"][1] * isz], np.uint64)
ptr = lambda a: a.ctypes.data_as(C.c_void_p)
n = c["nctu"] * 3 * 5 * 32
d_cnt = torch.zeros(n, dtype=torch.int32, device="cuda"); d_org = torch.zeros(n, dtype=torch.int32, device="cuda")
rt, ft = tab(d_rec), tab(d_fenc)
f = L.lib.x265amd_sao_stats_rows_cols
L.lib.x265amd_last_error.restype = C.c_char_p
def unit(r, c0, c1):
    rc = f(None, ptr(rt), ptr(ft), C.c_int64(c["stride"]), C.c_int64(c["cstride"]), w, h, C.c_void_p(d_cnt.data_ptr()), C.c_void_p(d_org.data_ptr()), r, r + 1, c0, c1)
    assert rc == 0, L.lib.x265amd_last_error()
def unit_old(r, c0, c1):
    assert f(None, ptr(rt), ptr(ft), C.c_int64(c["stride"]), C.c_int64(c["cstride"]), w, h, C.c_void_p(d_cnt.data_ptr()), C.c_void_p(d_org.data_ptr()), r, r + 1, c0, c1) == 0
def measure(tag, cols):
    for _ in range(20): unit(3, 4, 4 + cols)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    ts = []
    for k in range(200):
        e0.record(); unit(3 + k % 5, 4, 4 + cols); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1000)
    ts.sort()
    print("%s: %d CTUs x 3 planes per launch: median %.1f us, fastest %.1f us (events around one launch)" % (tag, cols, ts[len(ts) // 2], ts[0]))
measure("alone", 2); measure("alone", 26)
L.lib.x265amd_queue_acquire.restype = C.c_void_p
q = L.lib.x265amd_queue_acquire()
time.sleep(0.05)
measure("beside the resident job server", 2); measure("beside the resident job server", 26)
L.lib.x265amd_queue_release.argtypes = [C.c_void_p]
L.lib.x265amd_queue_release(q)
