# the lookahead's batch kernels beside the first pictures (hold 0) and alone (hold 1: every decision before the first picture starts): rocprofv3 kernel trace, 2160p
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for hold in 0 1; do
  X265AMD_HOLD_UNTIL_FLUSH=$hold X265AMD_TIMING=1 rocprofv3 --kernel-trace -d gpurun_out/la$hold -o la -- python3 dbg/enc_cfg.py 3840x2160 medium 8 20 ${WARM:-0} 2> gpurun_out/la_err_$hold.txt | tail -1
  grep "decision:\|lookahead:" gpurun_out/la_err_$hold.txt | head -4
  python3 - gpurun_out/la$hold/la_results.db <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
rows = list(db.execute("select * from kernels"))
ni, si, ei, gx, gy = (cols.index(c) for c in ("name", "start", "end", "grid_x", "grid_y"))
t0 = min(r[si] for r in rows)
for r in rows:
    if "lowres_cost" in r[ni] or "job_server" in r[ni]:
        print("  %-24s start %8.1f ms dur %8.2f ms grid %s x %s" % (r[ni][:24], (r[si] - t0) / 1e6, (r[ei] - r[si]) / 1e6, r[gx], r[gy]))
PY
done
