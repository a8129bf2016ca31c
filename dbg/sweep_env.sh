#!/bin/bash
# one environment variable over the bench's clips: dbg/sweep_env.sh <outdir> VAR v1 v2 ...   (two runs each; seconds per clip)
out=$1; var=$2; shift 2; mkdir -p $out; : > $out/sweep.txt
for v in "$@"; do
  for cfg in "1920x1080 medium 8 20" "3840x2160 medium 8 20" "1920x1080 medium 8 60" "3840x2160 medium 10 12" "3840x2160 slow 8 12"; do
    for rep in 1 2; do
      echo "$var=$v cfg $cfg rep $rep: $(env $var=$v timeout 200 python dbg/enc_cfg.py $cfg 2 2>/dev/null | tail -1)" >> $out/sweep.txt
    done
  done
done
awk '{print $1, $3, $4, $5, $6, $12, $NF}' $out/sweep.txt | awk '{k=$2" "$3" "$4" "$5; s[k]=s[k]"  "$1":"$6; m[k]=m[k]" "substr($7,1,6)} END{for(k in s) print k": "s[k]" |"m[k]}' | sort
