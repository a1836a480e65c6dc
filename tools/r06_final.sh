#!/bin/bash
# tools/r06_final.sh — what the round ends with, at the final tree: the -m gpu suite, smoke(), the driver's own bench command, and the
# multi-rank forms of bench.py typed without a launcher (rehearsals on the one GPU: gloo, 2 ... 5 ranks; one process; one rank through RCCL)
set -u
out=gpurun_out/r06f
mkdir -p $out
run() { local secs=$1 log=$2; shift 2; echo "=== $*"; timeout -k 10 "$secs" "$@" > "$log" 2> "${log%.log}.err"; local rc=$?; echo "=== rc=$rc $log"; [ $rc -eq 124 ] && { echo "timed out: stopping"; exit 124; }; return 0; }
run 1000 $out/gpu_tests.log python -X faulthandler -m pytest tests -m gpu -x -q --durations=8
tail -4 $out/gpu_tests.log
run 300 $out/smoke.log python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
tail -1 $out/smoke.log
run 400 $out/bench_final.log python bench.py --gpus 1 --steps 20 --warmup 5
for k in 2 3 4 5; do
  run 500 $out/bench_gpus${k}_as_typed_gloo.log python bench.py --gpus $k --backend gloo --share-gpu --steps 20 --warmup 5
done
run 400 $out/bench_gpus2_one_process.log python bench.py --gpus 2 --one-process --share-gpu --steps 20 --warmup 5
run 400 $out/bench_rehearse_sharded_rccl_1rank.log python bench.py --rehearse-sharded --steps 20 --warmup 5
echo done
