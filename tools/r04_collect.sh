#!/bin/bash
# tools/r04_collect.sh — the round-4 measurement set in one gpurun call: bench line, rocprofv3 kernel statistics of the same
# command, PMC passes for three workloads (headline, headline with individual masses, the reference's own 25 000-body
# workload size with individual masses), kernel statistics of the reference workload and of N = 65 536, enqueue depth.
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
run() { local secs=$1 log=$2; shift 2; echo "=== $*" ; timeout -k 10 "$secs" "$@" > "$log" 2>&1; local rc=$?; echo "=== rc=$rc $log"; [ $rc -eq 124 ] && { echo "timed out: stopping"; exit 124; }; return 0; }
run 400 gpurun_out/r04_bench.log python bench.py --steps 20 --warmup 3
tail -c 600 gpurun_out/r04_bench.log; echo
tools/gpu_round.sh prof
PMC_DIR=gpurun_out/pmc_head tools/gpu_round.sh pmc
PMC_DIR=gpurun_out/pmc_gen PMC_ARGS="--general-mass" tools/gpu_round.sh pmc
PMC_DIR=gpurun_out/pmc_25k PMC_ARGS="--nbodies 25000 --general-mass --steps 100 --warmup 5" tools/gpu_round.sh pmc
rm -rf gpurun_out/prof_ref gpurun_out/prof_64k
run 200 gpurun_out/r04_ref_workload_stats.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ref -- ./build/nbody_main -reference-ics -s 400
run 200 gpurun_out/r04_n65536_stats.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_64k -- ./build/nbody_main -n 65536 -s 200
run 120 gpurun_out/r04_enqueue_depth.log python tools/enqueue_depth.py --n 98304
run 120 gpurun_out/r04_enqueue_depth_allreduce.log python tools/enqueue_depth.py --n 98304 --allreduce
run 300 gpurun_out/r04_bench_general.log python bench.py --steps 20 --warmup 3 --no-cpu-baseline --general-mass --no-secondary
run 200 gpurun_out/r04_bench_25k.log python bench.py --steps 200 --warmup 20 --no-cpu-baseline --nbodies 25000 --general-mass --no-secondary
run 200 gpurun_out/r04_bench_64k.log python bench.py --steps 100 --warmup 10 --no-cpu-baseline --nbodies 65536 --no-secondary
echo done
