#!/usr/bin/env python3
"""Reduces profiles/collect_traffic.sh: HBM-side traffic of the job server's command kinds (the same device code launched as ordinary kernels, FETCH_SIZE and WRITE_SIZE in
separate rocprofv3 --pmc passes) against the algorithmic bytes the commands count for themselves -> traffic.json.
Units and corrections (MI355X_MICROARCH.md, HBM / rocprofv3): the counters are KB; on gfx950 FETCH_SIZE tallies 64 B per read request, so wide coalesced (128 B)
requests are under-reported by 2x -- these commands read 1-8 bytes per lane, whose requests are 64 B or less: the raw figure is the lower bound, twice it the upper;
WRITE_SIZE is uncalibrated (taken as it comes).  Infinity-Cache hits are counted as traffic, not excluded."""
import csv, glob, json, os, sys, collections
out, dst = sys.argv[1], sys.argv[2]
KIND = {"k_intra_nxn": "intra_nxn", "k_copy_rects": "copy_rects", "k_cu_measure": "cu_measure", "k_cu_measure_wg": "cu_measure", "k_tu_chain": "tu_chain", "k_intra_tu_chain": "intra_tu_chain", "k_intra_scan": "intra_scan",
        "k_intra_pu": "intra_pu", "k_motion_compensation": "mc", "k_mc_cost": "mc_cost", "k_me_search": "me_search", "k_me_deferred": "me_deferred", "k_est_bit": "est_bit"}
def find(sub, pat):
    g = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return g[0] if g else None
res = collections.defaultdict(lambda: {"launches": 0, "FETCH_SIZE_KB": 0.0, "WRITE_SIZE_KB": 0.0})
for sub, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = find(sub, "*counter_collection.csv")
    if not f: continue
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != ctr: continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].strip()
        res[name][ctr + "_KB"] += float(r["Counter_Value"])
        if ctr == "FETCH_SIZE": res[name]["launches"] += 1
cmds = json.load(open(os.path.join(out, "commands.json")))
rows = {}
for name, d in sorted(res.items(), key=lambda kv: -(kv[1]["FETCH_SIZE_KB"] + kv[1]["WRITE_SIZE_KB"])):
    kind = KIND.get(name)
    c = cmds["per_command_kind"].get(kind) if kind else None
    row = {"launches": d["launches"], "fetch_bytes_raw": int(d["FETCH_SIZE_KB"] * 1024), "write_bytes_raw": int(d["WRITE_SIZE_KB"] * 1024)}
    row["hbm_bytes_low"] = row["fetch_bytes_raw"] + row["write_bytes_raw"]
    row["hbm_bytes_high"] = 2 * row["fetch_bytes_raw"] + row["write_bytes_raw"]
    if c:
        row["command_kind"] = kind; row["commands_in_the_job_server_run"] = c["commands"]; row["algorithmic_bytes"] = c["algorithmic_bytes"]
        if c["algorithmic_bytes"]:
            row["traffic_over_algorithmic_low"] = round(row["hbm_bytes_low"] / c["algorithmic_bytes"], 3)
            row["traffic_over_algorithmic_high"] = round(row["hbm_bytes_high"] / c["algorithmic_bytes"], 3)
            # per unit of the kind: the two runs do not issue the same NUMBER of units of a kind once P / B pictures are in (with the job server the skip chain and the fused
            # search stand for most merge checks and searches, which as launches are motion compensations, measurements and transform chains of their own), so the totals
            # above compare like with like only for the kinds of an I picture; a launch's traffic against a command's own algorithmic bytes does for the others
            if d["launches"] and c["commands"]:
                row["hbm_bytes_per_launch_low"] = int(row["hbm_bytes_low"] / d["launches"]); row["hbm_bytes_per_launch_high"] = int(row["hbm_bytes_high"] / d["launches"])
                row["algorithmic_bytes_per_command"] = int(c["algorithmic_bytes"] / c["commands"])
                row["per_unit_traffic_over_algorithmic_low"] = round(row["hbm_bytes_per_launch_low"] / max(1, row["algorithmic_bytes_per_command"]), 3)
                row["per_unit_traffic_over_algorithmic_high"] = round(row["hbm_bytes_per_launch_high"] / max(1, row["algorithmic_bytes_per_command"]), 3)
    rows[name] = row
tot_low = sum(r["hbm_bytes_low"] for r in rows.values() if "command_kind" in r); tot_high = sum(r["hbm_bytes_high"] for r in rows.values() if "command_kind" in r)
tot_alg = sum(r["algorithmic_bytes"] for r in rows.values() if "command_kind" in r)
json.dump({"frames": cmds["frames"], "stream_md5": cmds["stream_md5"], "kernels": rows,
           "commands_total": {"hbm_bytes_low": tot_low, "hbm_bytes_high": tot_high, "algorithmic_bytes": tot_alg,
                              "traffic_over_algorithmic_low": round(tot_low / tot_alg, 3) if tot_alg else None, "traffic_over_algorithmic_high": round(tot_high / tot_alg, 3) if tot_alg else None},
           "note": __doc__}, open(dst, "w"), indent=1)
print(json.dumps({k: v for k, v in rows.items() if "command_kind" in v}, indent=1)[:3000])
