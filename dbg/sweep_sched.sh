#!/bin/bash
# scheduling switches over the bench's clips AND two long clips (an I picture every 24 frames): dbg/sweep_sched.sh <outdir> "A=1 B=2" "A=3" ...
out=$1; shift; mkdir -p $out; : > $out/sweep.txt
for v in "$@"; do
  for cfg in "1920x1080 medium 8 20" "3840x2160 medium 8 20" "1920x1080 medium 8 60" "3840x2160 slow 8 12"; do
    echo "[$v] cfg $cfg rep 1: $(env $v timeout 200 python dbg/enc_cfg.py $cfg 2 2>/dev/null | tail -1)" >> $out/sweep.txt
  done
  echo "[$v] cfg long1080x192 rep 1: $(env $v timeout 300 python dbg/long_clip.py 1920x1080 192 noref 2>/dev/null | tail -1 | sed 's/x265amd: \([0-9]*\) frames in \([0-9.]*\) s.*md5 \(.*\)/frames \1 seconds \2 fps 0 bytes 0 \3/')" >> $out/sweep.txt
  echo "[$v] cfg long2160x96 rep 1: $(env $v timeout 300 python dbg/long_clip.py 3840x2160 96 noref 2>/dev/null | tail -1 | sed 's/x265amd: \([0-9]*\) frames in \([0-9.]*\) s.*md5 \(.*\)/frames \1 seconds \2 fps 0 bytes 0 \3/')" >> $out/sweep.txt
done
python3 - $out/sweep.txt <<'PY'
import re, sys, collections
t = collections.OrderedDict()
for l in open(sys.argv[1]):
    m = re.match(r'\[(.*)\] cfg (.*) rep \d: frames \d+ seconds ([\d.]+) .* (\w+)$', l.strip())
    if m: t.setdefault(m.group(2), collections.OrderedDict()).setdefault(m.group(1), []).append((m.group(3), m.group(4)[:6]))
for cfg, d in t.items():
    print(cfg + ":  " + "   ".join("[%s] %s" % (k.replace("X265AMD_", ""), "/".join(s for s, _ in v)) for k, v in d.items()) + "   md5 " + ",".join(sorted(set(h for v in d.values() for _, h in v))))
PY
