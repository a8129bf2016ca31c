#!/bin/bash
# X265AMD_QUEUES sweep over the bench's three medium clips: dbg/sweep_queues.sh <outdir>
out=$1; mkdir -p $out
for q in 128 160 192 224; do
  for cfg in "1920x1080 medium 8 20" "3840x2160 medium 8 20" "1920x1080 medium 8 60"; do
    for rep in 1 2; do
      echo "queues $q cfg $cfg rep $rep: $(X265AMD_QUEUES=$q timeout 120 python dbg/enc_cfg.py $cfg 2 2>/dev/null | tail -1)" >> $out/sweep.txt
    done
  done
done
cat $out/sweep.txt
