import sys, time; sys.path.insert(0,'tests')
import numpy as np, hevc_testlib as T
g=np.load('tests/golden/frame_pipeline_golden.npz')
me=T.HipME(8); L=T.load_hip(8)
orig=L.lib.x265amd_analyse_frame
times=[]
class W:
    def __call__(self,*a):
        t=time.time(); r=orig(*a); times.append(time.time()-t); return r
L.lib.x265amd_analyse_frame=W()
for tag in ("deblock/","bframes/"):
    times.clear()
    sched=g[tag+"schedule"]
    t0=time.time()
    T.frame_pipeline_run_hip(L,me,[int(q) for q in g[tag+"slice_qp"]],nframes=len(sched),deblock=True,schedule=sched,frames=T.frame_clip_b(8) if "bframes" in tag else None)
    print(tag,"total %.2fs"%(time.time()-t0),"analyse per frame:",["%.2f"%t for t in times],"types",sched[:,0].tolist())
