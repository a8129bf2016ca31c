for v in "X265AMD_SPIN_US=0" "X265AMD_SPIN_US=30" "X265AMD_SPIN_US=60" "X265AMD_SPIN_US=60 X265AMD_SPIN_ROWS=3" "X265AMD_SPIN_US=60 X265AMD_SPIN_ROWS=17" "X265AMD_SPIN_US=200"; do for r in 1 2; do
echo "$v 24: $(env $v timeout 300 python dbg/enc_bench.py 24 2 2>/dev/null | tail -1 | cut -d' ' -f1-6)"
done; echo "$v 60: $(env $v timeout 300 python dbg/enc_clip60.py 2>/dev/null | tail -1| cut -d' ' -f1-6)"; done
