# X265AMD_QUEUE_LOG=poc,row: summary of the row's command / wait timeline (dbg/qlog_report.py log [first last])
import sys, re
names = "NOP EXIT COPY COPY2D FILL COPY_RECTS MC MC_COST CU_MEASURE TU_CHAIN TU_CHAIN_RDOQ INTRA_TU_CHAIN INTRA_TU_CHAIN_RDOQ INTRA_SCAN ME_SEARCH ME_SEARCH_STAR ME_DEFERRED EST_BIT INTRA_PU INTRA_NXN INTER_CHAIN INTER_SEARCH WAIT".split()
ev = []
for l in open(sys.argv[1]):
    m = re.match(r"x265amd qlog poc (\d+) row (\d+): ([\d.]+) ([\d.]+) (\w) (\d+)", l)
    if m: ev.append((float(m.group(3)), float(m.group(4)), m.group(5), int(m.group(6))))
if not ev: sys.exit("no events")
t0 = ev[0][0]
tot_wait = tot_run = tot_off = 0.0; waits = 0
print("events", len(ev), "span %.1f us" % (ev[-1][0] - t0))
# between events: wall delta, run delta (host CPU of the task), and what is neither (parked / waiting for a worker)
pend = []
rows = []
for i in range(1, len(ev)):
    w = ev[i][0] - ev[i - 1][0]; r = ev[i][1] - ev[i - 1][1]
    if ev[i][2] == 'R' and ev[i - 1][2] == 'W':
        tot_wait += w; waits += 1
        rows.append(("wait", w, r, [names[o] for o in pend])); pend = []
    else:
        tot_run += r; tot_off += w - r
        if ev[i][2] == 'E': pend.append(ev[i][3])
        rows.append((ev[i][2], w, r, []))
print("waits %d: %.1f us in all (%.1f each); host running %.1f us; not running outside waits %.1f us" % (waits, tot_wait, tot_wait / max(1, waits), tot_run, tot_off))
a = int(sys.argv[2]) if len(sys.argv) > 2 else 0; b = int(sys.argv[3]) if len(sys.argv) > 3 else 0
acc = 0.0; accr = 0.0
k = 0
for kind, w, r, ops in rows:
    if kind == "wait":
        if a <= k < b: print("  host %.1f us (running %.1f) then wait %.1f us for %s" % (acc, accr, w, " ".join(ops)))
        acc = accr = 0.0; k += 1
    else:
        acc += w; accr += r
# histogram of waits by command set
from collections import defaultdict
h = defaultdict(lambda: [0, 0.0])
for kind, w, r, ops in rows:
    if kind == "wait": x = h[" ".join(ops)]; x[0] += 1; x[1] += w
for key, (n, t) in sorted(h.items(), key=lambda kv: -kv[1][1])[:12]:
    print("  %5d x %7.1f us  %s" % (n, t / n, key))
