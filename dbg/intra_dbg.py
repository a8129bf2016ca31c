import sys; sys.path.insert(0,'tests')
import numpy as np, hevc_testlib as T, test_intra_rd as M
gold=np.load(M.GOLD_PATH)
for k,(depth,seed,st,psy,strong) in enumerate(M.CASES[:3]):
    c=T.intra_rd_case(depth,seed,st,psy,strong=strong)
    got=T.intra_rd_pack(T.intra_rd_run_hip(T.load_hip(depth),c),c)
    for i,d in enumerate(got):
        bad=[n for n in ("info","dirs","pred","units","coeff","recon","res","ctx") if not np.array_equal(np.asarray(d[n]),gold["%d/%d/%s"%(k,i,n)])]
        cu=c["cus"][i]; S=1<<int(cu["log2_size"])
        msg=""
        if "coeff" in bad:
            a=np.asarray(d["coeff"]); w=gold["%d/%d/coeff"%(k,i)]
            idx=np.nonzero(a!=w)[0]
            msg=" coeff first bad %d (luma %d, U to %d) n=%d"%(idx[0],S*S,S*S+S*S//4,len(idx))
        if "res" in bad: msg+=" res got %s want %s"%(d["res"].tolist(),gold["%d/%d/res"%(k,i)].tolist())
        print(k,i,"x",cu["x"],"y",cu["y"],"log2",cu["log2_size"],"ldir",d["dirs"][0,0],"cdir",d["dirs"][0,1],"want cdir",gold["%d/%d/dirs"%(k,i)][0,1],"bad",bad,msg)
