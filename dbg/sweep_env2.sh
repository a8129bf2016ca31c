#!/bin/bash
# settings of several environment variables over the bench's clips: dbg/sweep_env2.sh <outdir> "A=1 B=2" "A=3 B=4" ...   (two runs each; seconds per clip)
out=$1; shift; mkdir -p $out; : > $out/sweep.txt
for v in "$@"; do
  for cfg in "1920x1080 medium 8 20" "3840x2160 medium 8 20" "1920x1080 medium 8 60" "3840x2160 medium 10 12" "3840x2160 slow 8 12"; do
    for rep in 1 2; do
      echo "[$v] cfg $cfg rep $rep: $(env $v timeout 200 python dbg/enc_cfg.py $cfg 2 2>/dev/null | tail -1)" >> $out/sweep.txt
    done
  done
done
python3 - $out/sweep.txt <<'PY'
import re, sys, collections
t = collections.OrderedDict()
for l in open(sys.argv[1]):
    m = re.match(r'\[(.*)\] cfg (.*) rep \d: frames \d+ seconds ([\d.]+) .* (\w+)$', l.strip())
    if m: t.setdefault(m.group(2), collections.OrderedDict()).setdefault(m.group(1), []).append((m.group(3), m.group(4)[:6]))
for cfg, d in t.items():
    print(cfg + ":  " + "   ".join("[%s] %s" % (k, "/".join(s for s, _ in v)) for k, v in d.items()) + "   md5 " + ",".join(sorted(set(h for v in d.values() for _, h in v))))
PY
