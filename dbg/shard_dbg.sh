#!/bin/bash
# the two-process one-GPU shard leg of the bench with timing output per rank (tests/test_bench_shard_frames.py's command): dbg/shard_dbg.sh <outdir> [runs]
out=${1:-gpurun_out/shard}; runs=${2:-3}; mkdir -p $out
for i in $(seq 1 $runs); do
  X265AMD_IMPORT_WAIT_S=25 X265AMD_TIMING=1 X265AMD_QUEUES=${QUEUES:-64} MASTER_ADDR=127.0.0.1 timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29500 + i)) bench.py --gpus 2 --steps 10 --warmup 0 --shard frames --backend gloo --one-gpu --no-kernel-workload --no-cpu-baseline > $out/run$i.out 2> $out/run$i.err
  echo "run $i rc $?: $(grep -c 'handed to a frame task' $out/run$i.err) hand-overs, $(grep -c 'poc [0-9]* type' $out/run$i.err) pictures coded, $(grep -c 'did not arrive\|failed' $out/run$i.err) failures"
done
