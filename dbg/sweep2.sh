for v in "X265AMD_EARLY_BREF=0" "X265AMD_EARLY_BREF=1" "X265AMD_LAZY_MC=0" "X265AMD_EARLY_P_MAX=12" "X265AMD_FRAME_THREADS=14" "X265AMD_FRAME_THREADS=14 X265AMD_EARLY_BREF=0"; do for r in 1 2; do
echo "$v 24: $(env $v timeout 300 python dbg/enc_bench.py 24 2 2>/dev/null | tail -1 | cut -d' ' -f1-6)"
done; echo "$v 60: $(env $v timeout 300 python dbg/enc_clip60.py 2>/dev/null | tail -1| cut -d' ' -f1-6)"; done
