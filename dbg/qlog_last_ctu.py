# X265AMD_QUEUE_LOG=poc,row: the waits of the row's last CTU (the events behind the last long pause of the row's queue): dbg/qlog_last_ctu.py log
import re, sys
names = "NOP EXIT COPY COPY2D FILL COPY_RECTS MC MC_COST CU_MEASURE TU_CHAIN TU_CHAIN_RDOQ INTRA_TU_CHAIN INTRA_TU_CHAIN_RDOQ INTRA_SCAN ME_SEARCH ME_SEARCH_STAR ME_DEFERRED EST_BIT INTRA_PU INTRA_NXN INTER_CHAIN INTER_SEARCH WAIT".split()
ev = []
for l in open(sys.argv[1]):
    m = re.match(r"x265amd qlog poc (\d+) row (\d+): ([\d.]+) ([\d.]+) (\w) (\d+)", l)
    if m: ev.append((float(m.group(3)), m.group(5), int(m.group(6))))
start = 0
for i in range(len(ev) - 1, 0, -1):
    if ev[i][0] - ev[i - 1][0] > 400 and ev[i][1] == 'E': start = i; break
print("last CTU: %.1f us" % (ev[-1][0] - ev[start][0]))
pend = []; last = ev[start][0]; w0 = last; tw = th = 0
for t, k, o in ev[start:]:
    if k == 'E': pend.append(names[o])
    elif k == 'W': w0 = t
    elif k == 'R':
        print("host %6.1f  wait %6.1f  %s" % (w0 - last, t - w0, " ".join(pend))); th += w0 - last; tw += t - w0; pend = []; last = t
print("host %.1f us, waits %.1f us" % (th, tw))
