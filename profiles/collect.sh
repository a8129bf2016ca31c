#!/usr/bin/env bash
# Runs on the GPU box (gpurun): the bench line, rocprofv3 kernel statistics of the same command (the encoder: one resident job-server kernel plus the
# in-loop filter kernels; the kernel workload: the batched hot-path kernels), and separate FETCH_SIZE / WRITE_SIZE passes over the kernel workload alone
# (bench_kernels.py: counter collection serialises kernel dispatches, which a resident kernel next to ordinary launches does not survive),
# reduced to profiles-style files under gpurun_out/prof_<tag>/.  usage: bash profiles/collect.sh <tag> [bench args]
set -u
TAG=${1:-final}; shift || true
ROUND=${ROUND:-r04}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 900 python3 $ROOT/bench.py "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > "$OUT/kt.log" 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 $ROOT/bench_kernels.py --steps 5 --warmup 1 --no-cpu-baseline > "$OUT/fetch.log" 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 $ROOT/bench_kernels.py --steps 5 --warmup 1 --no-cpu-baseline > "$OUT/write.log" 2>&1
python3 $ROOT/profiles/reduce.py "$OUT" "$TAG" "$ROUND"
