#!/bin/bash
# A/B of an environment switch over the bench's clips: dbg/ab.sh <outdir> VAR valueA valueB
out=$1; var=$2; mkdir -p $out
for rep in 1 2 3; do
  for v in $3 $4; do
    for cfg in "1920x1080 medium 8 20" "3840x2160 medium 8 20" "1920x1080 medium 8 60"; do
      echo "$var=$v cfg $cfg rep $rep: $(env $var=$v timeout 120 python dbg/enc_cfg.py $cfg 2 2>/dev/null | tail -1)" >> $out/ab.txt
    done
  done
done
cat $out/ab.txt
