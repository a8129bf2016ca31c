# row ends of chosen pictures from an X265AMD_TIMING=1 log (the timed encode): dbg/rows_of.py log poc [poc ...]
import re, sys, collections
lines = open(sys.argv[1]).read().splitlines()
if "---- timed encode ----" in lines: lines = lines[len(lines) - 1 - lines[::-1].index("---- timed encode ----"):]
by = collections.defaultdict(dict)
t0 = None
for l in lines:
    m = re.match(r'x265amd: row: poc (\d+) row (\d+) start ([\d.]+) queue ([\d.]+) end ([\d.]+) ran ([\d.]+)', l)
    if m:
        p, r = int(m.group(1)), int(m.group(2)); s, q, e, ran = (float(m.group(k)) for k in (3, 4, 5, 6))
        by[p][r] = (s, q, e, ran); t0 = s if t0 is None else min(t0, s)
pocs = [int(a) for a in sys.argv[2:]]
print("row " + " ".join("| poc %2d start   end   ran " % p for p in pocs))
for r in range(max(len(by[p]) for p in pocs)):
    print("%3d " % r + " ".join("| %13.1f %6.1f %5.1f" % (by[p][r][1] - t0, by[p][r][2] - t0, by[p][r][3]) if r in by[p] else "|" + " " * 26 for p in pocs))
