#!/usr/bin/env python3
"""Reduces the rocprofv3 outputs of profiles/collect.sh: kernel_stats.csv (copied), per-kernel mean FETCH_SIZE/WRITE_SIZE per
launch -> traffic.json.  Counters are raw KB as rocprofv3 reports them (MI355X_MICROARCH.md 'HBM': FETCH_SIZE =
TCC_EA0_RDREQ x 64 B on gfx950, i.e. it under-reports wide 128-B requests by up to 2x; WRITE_SIZE uncalibrated)."""
import csv, glob, json, os, shutil, sys, collections
out, tag = sys.argv[1], sys.argv[2]
rnd = sys.argv[3] if len(sys.argv) > 3 else "r01"
def find(sub, pat):
    g = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return max(g, key=os.path.getmtime) if g else None          # (gpurun_out/ keeps earlier collections' files: the newest is this one's)
ks = find("kt", "*kernel_stats.csv")
if ks: shutil.copy(ks, os.path.join(out, "%s_%s_kernel_stats.csv" % (rnd, tag)))
res = collections.defaultdict(dict)
for sub, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = find(sub, "*counter_collection.csv")
    if not f: continue
    shutil.copy(f, os.path.join(out, "%s_%s_pmc_%s.csv" % (rnd, tag, sub)))
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == ctr: acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        name = k.split("(")[0].replace("void ", "").split("<")[0]
        res[name][ctr + "_KB_per_launch"] = res[name].get(ctr + "_KB_per_launch", 0) + sum(v) / len(v)
for k, d in res.items():
    d["hbm_bytes_per_launch_uncorrected"] = int((d.get("FETCH_SIZE_KB_per_launch", 0) + d.get("WRITE_SIZE_KB_per_launch", 0)) * 1024)
res["note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over the kernel workload (`bench_kernels.py --steps 5`); raw counters in KB, mean per launch. "
               "gfx950: FETCH_SIZE counts 64 B per read request, so wide (128 B) coalesced requests are under-reported by up to 2x; "
               "these kernels issue 1-4 B/lane loads, so the read side lies between 1x and 2x of the raw figure (MI355X_MICROARCH.md, HBM section).")
json.dump(res, open(os.path.join(out, "%s_%s_traffic.json" % (rnd, tag)), "w"), indent=1)
print(open(os.path.join(out, "bench.json")).read().strip()[-1500:])
print(json.dumps(res, indent=1)[:1500])
