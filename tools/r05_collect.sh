#!/bin/bash
# tools/r05_collect.sh — the round-5 measurement set in one gpurun call:
#   1. bench line + rocprofv3 kernel statistics of the same command (headline, equal masses)
#   2. PMC passes: headline; headline with individual masses (12 + 2 body, --general-mass = forced off / default = measured);
#      the reference's own workload size (N = 25 000, individual masses); the fp64 kernel (config 5's size)
#   3. kernel statistics of the reference's default workload through the C host, with the library's default (mass folding by
#      measurement) and with the per-pair multiplies forced; N = 65 536; fp64
#   4. the per-workgroup timeline of the product kernel at N = 25 000 (12 + 2 and 11 + 2 bodies)
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
run() { local secs=$1 log=$2; shift 2; echo "=== $*" ; timeout -k 10 "$secs" "$@" > "$log" 2>&1; local rc=$?; echo "=== rc=$rc $log"; [ $rc -eq 124 ] && { echo "timed out: stopping"; exit 124; }; return 0; }
run 400 gpurun_out/r05_bench.log python bench.py --steps 20 --warmup 3
tail -c 400 gpurun_out/r05_bench.log; echo
tools/gpu_round.sh prof
PMC_DIR=gpurun_out/pmc_head tools/gpu_round.sh pmc
PMC_DIR=gpurun_out/pmc_gen PMC_ARGS="--general-mass --mass-scaling off" tools/gpu_round.sh pmc
PMC_DIR=gpurun_out/pmc_gen_default PMC_ARGS="--general-mass" tools/gpu_round.sh pmc
PMC_DIR=gpurun_out/pmc_25k PMC_ARGS="--nbodies 25000 --general-mass --mass-scaling off --steps 100 --warmup 5" tools/gpu_round.sh pmc
PMC_DIR=gpurun_out/pmc_fp64 PMC_ARGS="--precision fp64" tools/gpu_round.sh pmc
tools/gpu_round.sh prof64
rm -rf gpurun_out/prof_ref gpurun_out/prof_ref_unscaled gpurun_out/prof_64k
run 200 gpurun_out/r05_ref_workload_stats.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ref -- ./build/nbody_main -reference-ics -s 400
run 200 gpurun_out/r05_ref_workload_unscaled_stats.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ref_unscaled -- ./build/nbody_main -reference-ics -s 400 -no-mass-scaling
run 200 gpurun_out/r05_n65536_stats.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_64k -- ./build/nbody_main -n 65536 -s 200
run 120 gpurun_out/r05_ref_workload_run.log ./build/nbody_main -reference-ics -s 2000
run 120 gpurun_out/r05_ref_workload_unscaled_run.log ./build/nbody_main -reference-ics -s 2000 -no-mass-scaling
{ ./build/sym_timeline 25000 512 0 1 0 20; ./build/sym_timeline 25000 512 0 2 0 20; ./build/sym_timeline 25000 512 0 0 0 20; } > gpurun_out/r05_sym_timeline.log 2>&1
run 300 gpurun_out/r05_bench_general.log python bench.py --steps 20 --warmup 3 --no-cpu-baseline --general-mass --mass-scaling off --no-secondary
run 300 gpurun_out/r05_bench_general_default.log python bench.py --steps 20 --warmup 3 --no-cpu-baseline --general-mass --no-secondary
run 200 gpurun_out/r05_bench_25k.log python bench.py --steps 200 --warmup 20 --no-cpu-baseline --nbodies 25000 --general-mass --mass-scaling off --no-secondary
run 300 gpurun_out/r05_bench_fp64.log python bench.py --steps 20 --warmup 3 --no-cpu-baseline --precision fp64
echo done
