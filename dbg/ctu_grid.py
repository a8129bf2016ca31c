# X265AMD_CTU_LOG=p<poc>: the picture's CTUs as a grid of (gate wait, analysis) milliseconds and the chain that ends last: dbg/ctu_grid.py log poc
import re, sys
poc = int(sys.argv[2]); g = {}
for l in open(sys.argv[1]):
    m = re.search(r"x265amd ctu: poc (\d+) row (\d+) col (\d+): row-above wait from ([\d.]+), gate passed ([\d.]+), done ([\d.]+)(?:; in the CTU: running ([\d.]+), waiting for the device ([\d.]+), for reference samples ([\d.]+))?", l)
    if m and int(m.group(1)) == poc: g[(int(m.group(2)), int(m.group(3)))] = tuple(float(m.group(k) or 0) for k in (4, 5, 6, 7, 8, 9))
if not g: sys.exit("no CTUs of that picture")
R = 1 + max(r for r, c in g); C = 1 + max(c for r, c in g)
for r in range(R):
    for c in range(C): g.setdefault((r, c), (0.0,) * 6)       # a line torn by another thread's output
t0 = min(v[0] for v in g.values())
print("picture %d: %d x %d CTUs, first CTU begins at %.1f, last ends %.1f ms later" % (poc, C, R, t0, max(v[2] for v in g.values()) - t0))
print("analysis ms per CTU (gate passed -> done):")
for r in range(R): print("%2d " % r + " ".join("%4.1f" % (g[(r, c)][2] - g[(r, c)][1]) for c in range(C)) + " | sum %5.1f  ends %6.1f" % (sum(g[(r, c)][2] - g[(r, c)][1] for c in range(C)), g[(r, C - 1)][2] - t0))
print("waiting at the reference gate, ms per CTU:")
for r in range(R): print("%2d " % r + " ".join("%4.1f" % (g[(r, c)][1] - g[(r, c)][0]) for c in range(C)))
# idle between the previous CTU of the row ending and this one's row-above wait ending = waiting for the row above
print("waiting for the row above (from the previous CTU's end to the gate), ms per CTU:")
for r in range(R): print("%2d " % r + " ".join("%4.1f" % (g[(r, c)][0] - (g[(r, c - 1)][2] if c else g[(r, c)][0])) for c in range(C)))
print("in the CTU: waiting for reference samples (a command's exact reach), ms:")
for r in range(R): print("%2d " % r + " ".join("%4.1f" % g[(r, c)][5] for c in range(C)))
print("in the CTU: waiting for the device, ms:")
for r in range(R): print("%2d " % r + " ".join("%4.1f" % g[(r, c)][4] for c in range(C)))
print("in the CTU: the host running, ms:")
for r in range(R): print("%2d " % r + " ".join("%4.1f" % g[(r, c)][3] for c in range(C)))
