#!/bin/bash
# two builds of dbg/rdoq_lat.hip (dbg/bin/rdoq_lat_v1: against the tu_dev.h of another commit, dbg/bin/rdoq_lat: against the working tree's) over a few shapes of
# input: the checksums must agree line by line.
#   git archive <commit> x265-amod_amd/csrc include | tar -x -C /tmp/old
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DX265AMD_DEPTH=8 -I/tmp/old/include -I/tmp/old/x265-amod_amd/csrc dbg/rdoq_lat.hip -o dbg/bin/rdoq_lat_v1
out=${1:-gpurun_out/rdoq_lat}; mkdir -p $out
for args in "8 4 1 2 0 30" "8 2 5 2 0 27" "8 2 25 1 0 22" "8 2 3 2 1 33" "8 2 0.4 2 0 37" "8 2 12 2 2 24"; do
  tag=$(echo $args | tr ' .' '__')
  timeout 300 dbg/bin/rdoq_lat_v1 $args > $out/v1_$tag.txt 2>&1
  timeout 300 dbg/bin/rdoq_lat $args > $out/new_$tag.txt 2>&1
  if diff <(grep -o "checksum [0-9a-f]*" $out/v1_$tag.txt) <(grep -o "checksum [0-9a-f]*" $out/new_$tag.txt) > /dev/null; then echo "args $args: checksums agree"; else echo "args $args: CHECKSUMS DIFFER"; fi
done
paste -d'\n' <(grep "us per call" $out/v1_8_4_1_2_0_30.txt | sed 's/^/v1  /') <(grep "us per call" $out/new_8_4_1_2_0_30.txt | sed 's/^/new /')
grep -A1 "us per call" $out/new_8_4_1_2_0_30.txt | grep cycles
