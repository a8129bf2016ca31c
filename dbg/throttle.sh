#!/bin/bash
# how often the cgroup's CPU quota froze the process during an encode (cpu.stat: nr_throttled / throttled_usec): dbg/throttle.sh <enc_cfg args...>
s0=$(cat /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' ')
python dbg/enc_cfg.py "$@" 2>/dev/null | tail -1
s1=$(cat /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' ')
python3 - "$s0" "$s1" <<'PY'
import sys
def p(s):
    t = s.split(); return {t[i]: int(t[i + 1]) for i in range(0, len(t) - 1, 2)}
a, b = p(sys.argv[1]), p(sys.argv[2])
print("  cpu.stat during the run (warm-up + timed encode): " + ", ".join("%s +%d" % (k, b[k] - a[k]) for k in ("usage_usec", "nr_periods", "nr_throttled", "throttled_usec") if k in a and k in b))
PY
