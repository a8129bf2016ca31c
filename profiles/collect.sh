#!/usr/bin/env bash
# Runs on the GPU box (gpurun): bench line + rocprofv3 kernel stats + separate FETCH_SIZE / WRITE_SIZE passes for the
# same command, reduced to profiles-style files under gpurun_out/prof_<tag>/.  usage: bash profiles/collect.sh <tag> [bench args]
set -u
TAG=${1:-final}; shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > "$OUT/kt.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > "$OUT/write.log" 2>&1
python3 $ROOT/profiles/reduce.py "$OUT" "$TAG"
