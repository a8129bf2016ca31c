#!/bin/bash
# X265AMD_QUEUES_FIRST sweep (the job server's first launch) over the bench's clips: dbg/sweep_first.sh <outdir>
out=$1; mkdir -p $out
for q in 224 96 64 128; do
  for cfg in "1920x1080 medium 8 20" "3840x2160 medium 8 20" "1920x1080 medium 8 60" "3840x2160 medium 10 12"; do
    for rep in 1 2; do
      echo "first $q cfg $cfg rep $rep: $(X265AMD_QUEUES_FIRST=$q X265AMD_TIMING=1 timeout 120 python dbg/enc_cfg.py $cfg 2 2>$out/err.txt | tail -1) | $(grep -a 'decision:' $out/err.txt | sort -t: -k3 -n | tail -1 | cut -c1-40)" >> $out/sweep.txt
    done
  done
done
cat $out/sweep.txt
