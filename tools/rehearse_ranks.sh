#!/bin/bash
# tools/rehearse_ranks.sh <ranks> [bench.py arguments ...] — the N > 1 path of bench.py as the driver launches it
# (torch.distributed.run, one process per rank), rehearsed on ONE GPU over gloo (RCCL refuses two ranks per device).
# Not a scaling measurement: it shows the line a node run prints (parity_check, safe_first, protocol_tuning, phases).
set -u
ranks=$1; shift
mkdir -p gpurun_out
log=gpurun_out/r05_bench_rehearsal_gloo_${ranks}ranks.log
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$ranks" --master-addr 127.0.0.1 --master-port $((29500 + ranks)) \
    bench.py --gpus "$ranks" --backend gloo --share-gpu --steps 10 --warmup 3 "$@" > "$log" 2>&1
rc=$?
echo "=== rehearsal with $ranks ranks: rc=$rc $log"
grep -c '^{' "$log"
exit $rc
