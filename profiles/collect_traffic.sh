#!/usr/bin/env bash
# Counter evidence for the job server's commands (runs on the GPU box): the I picture of the bench clip is coded once through the job server (per command kind: count,
# body time, algorithmic bytes) and then with X265AMD_QUEUES=0 -- every command an ordinary launch of the same device code -- under rocprofv3 --pmc FETCH_SIZE and, in a
# pass of its own, --pmc WRITE_SIZE (the counter passes serialise dispatches, which the resident kernel does not survive; the program directly behind `--`).
# usage: bash profiles/collect_traffic.sh [frames=1]   -> gpurun_out/traffic/{commands.json, fetch/, write/}; profiles/reduce_traffic.py reduces them
set -u
N=${1:-1}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/traffic
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 python3 $ROOT/dbg/enc_traffic.py $N "$OUT/commands.json" > "$OUT/queues.log" 2>&1
export X265AMD_QUEUES=0
timeout 900 python3 $ROOT/dbg/enc_traffic.py $N > "$OUT/launches.log" 2>&1
timeout 1500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 $ROOT/dbg/enc_traffic.py $N > "$OUT/fetch.log" 2>&1
timeout 1500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 $ROOT/dbg/enc_traffic.py $N > "$OUT/write.log" 2>&1
python3 $ROOT/profiles/reduce_traffic.py "$OUT" "$OUT/traffic.json"
tail -2 "$OUT/queues.log" "$OUT/launches.log" "$OUT/fetch.log" "$OUT/write.log"
