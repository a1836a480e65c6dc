#!/bin/bash
# tools/r06_collect.sh — the round-6 measurement set in one gpurun call (everything at the final tree):
#   1. bench line + rocprofv3 kernel statistics of the same command (headline, equal masses)
#   2. PMC passes: headline; headline with individual masses (12 + 2 body: the default since ABI 6); the reference's own
#      workload size (N = 25 000, individual masses); the fp64 kernel (config 5's size)
#   3. kernel statistics of the reference's default workload through the C host; N = 65 536; fp64
#   4. what the reference's caller pays per frame (C++ adaptor: sim_thread_example frames; bench.py's frame_ms)
#   5. `bench.py --gpus N` as typed (no launcher): 2 ... 5 ranks over gloo on the one GPU; the one-process fallback; the one-rank RCCL rehearsal
set -u
out=gpurun_out/r06c
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
run() { local secs=$1 log=$2; shift 2; echo "=== $*" ; timeout -k 10 "$secs" "$@" > "$log" 2> "${log%.log}.err"; local rc=$?; echo "=== rc=$rc $log"; [ $rc -eq 124 ] && { echo "timed out: stopping"; exit 124; }; return 0; }
run 400 $out/bench.log python bench.py --steps 20 --warmup 5
tail -c 300 $out/bench.log; echo
tools/gpu_round.sh prof
PMC_DIR=gpurun_out/pmc_head tools/gpu_round.sh pmc
PMC_DIR=gpurun_out/pmc_gen PMC_ARGS="--general-mass" tools/gpu_round.sh pmc
PMC_DIR=gpurun_out/pmc_25k PMC_ARGS="--nbodies 25000 --general-mass --steps 100 --warmup 5" tools/gpu_round.sh pmc
PMC_DIR=gpurun_out/pmc_fp64 PMC_ARGS="--precision fp64" tools/gpu_round.sh pmc
tools/gpu_round.sh prof64
rm -rf gpurun_out/prof_ref gpurun_out/prof_64k
run 200 $out/ref_workload_stats.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ref -- ./build/nbody_main -reference-ics -s 1000
run 200 $out/n65536_stats.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_64k -- ./build/nbody_main -n 65536 -s 200
run 120 $out/ref_workload_run.log ./build/nbody_main -reference-ics -s 2000
for i in 1 2 3; do ./build/sim_thread_example frames reference 300; done > $out/frames.log 2>&1
./build/sim_thread_example frames 262144 40 >> $out/frames.log 2>&1
run 300 $out/bench_general.log python bench.py --steps 20 --warmup 5 --no-cpu-baseline --general-mass --no-secondary
run 200 $out/bench_25k.log python bench.py --steps 200 --warmup 20 --no-cpu-baseline --nbodies 25000 --general-mass
run 300 $out/bench_fp64.log python bench.py --steps 20 --warmup 5 --no-cpu-baseline --precision fp64
for k in 2 3 4 5; do
  run 500 $out/bench_gpus${k}_as_typed_gloo.log python bench.py --gpus $k --backend gloo --share-gpu --steps 20 --warmup 5
done
run 400 $out/bench_gpus2_one_process.log python bench.py --gpus 2 --one-process --share-gpu --steps 20 --warmup 5
run 400 $out/bench_rehearse_sharded_rccl_1rank.log python bench.py --rehearse-sharded --steps 20 --warmup 5
echo done
