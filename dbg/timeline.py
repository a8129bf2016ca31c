# per-picture timeline of the last encode in an X265AMD_TIMING=1 log: dbg/timeline.py log
import re, sys, collections
rows = []
lines = open(sys.argv[1]).read().splitlines()
if "---- timed encode ----" in lines: lines = lines[len(lines) - 1 - lines[::-1].index("---- timed encode ----"):]      # dbg/enc_bench.py, enc_clip60.py: only the timed encode
for l in lines:
    m = re.match(r'x265amd: row: poc (\d+) row (\d+) start ([\d.]+) queue ([\d.]+) end ([\d.]+) ran ([\d.]+)', l)
    if m: rows.append(tuple(float(x) for x in m.groups()))
e = rows
by = collections.defaultdict(list)
for r in e: by[int(r[0])].append(r)
t0 = min(r[2] for r in e)
for poc in sorted(by):
    rs = sorted(by[poc], key=lambda r: r[1])
    s = min(r[2] for r in rs); en = max(r[4] for r in rs); lr = rs[-1]
    print("poc %2d start %6.1f end %6.1f | row0 %6.1f..%6.1f | last row %6.1f..%6.1f ran %5.1f | host ms all rows %6.1f" % (poc, s - t0, en - t0, rs[0][2] - t0, rs[0][4] - t0, lr[2] - t0, lr[4] - t0, lr[5], sum(r[5] for r in rs)))
print("span %.1f ms" % (max(r[4] for r in e) - t0))
