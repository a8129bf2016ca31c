for ft in 3 5 8 14; do for v in 0 1; do for r in 1 2; do
echo "FT=$ft EP=$v 24: $(X265AMD_FRAME_THREADS=$ft X265AMD_EARLY_P=$v timeout 300 python dbg/enc_bench.py 24 2 2>/dev/null | tail -1 | cut -d' ' -f1-6)"
echo "FT=$ft EP=$v 60: $(X265AMD_FRAME_THREADS=$ft X265AMD_EARLY_P=$v timeout 300 python dbg/enc_clip60.py 2>/dev/null | tail -1| cut -d' ' -f1-6)"
done; done; done
