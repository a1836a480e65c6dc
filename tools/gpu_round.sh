#!/bin/bash
# tools/gpu_round.sh — what one gpurun call runs: GPU tests, smoke, bench, (optional) profile.
# Ordinary failures continue to the next stage; a stage killed by its timeout stops the run
# (a hung GPU must not be poked again).
set -u
export PYTHONFAULTHANDLER=1   # a process abort (SIGABRT / SIGSEGV) inside the library leaves a Python traceback in the stage's log
mkdir -p gpurun_out
stage() {  # stage <seconds> <logfile> <cmd...>
    local secs=$1 log=$2; shift 2
    echo "=== $* (limit ${secs}s)" | tee -a gpurun_out/round.log
    timeout -k 10 "$secs" "$@" > "$log" 2>&1
    local rc=$?
    echo "=== rc=$rc $log" | tee -a gpurun_out/round.log
    # a failing stage leaves its whole log behind under a name of its own (copy it to profiles/ and commit it)
    if [ $rc -ne 0 ]; then cp "$log" "gpurun_out/FAILED_$(basename "$log" .log)_$(date -u +%H%M%S).log"; fi
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "stage timed out: stopping" | tee -a gpurun_out/round.log; tail -20 "$log"; exit $rc; fi
    return $rc
}
: > gpurun_out/round.log
for what in "$@"; do
  case $what in
    tests)  stage 900 gpurun_out/test_gpu.log python -X faulthandler -m pytest tests -m gpu -x -q --durations=8; tail -15 gpurun_out/test_gpu.log ;;
    testsall) stage 1100 gpurun_out/test_gpu.log python -X faulthandler -m pytest tests -m gpu -q --durations=12; tail -40 gpurun_out/test_gpu.log ;;
    smoke)  stage 300 gpurun_out/smoke.log python -c "import __graft_entry__ as g; g.smoke()"; tail -3 gpurun_out/smoke.log ;;
    bench)  stage 600 gpurun_out/bench.log python bench.py --steps 20 --warmup 3; tail -2 gpurun_out/bench.log ;;
    benchq) stage 300 gpurun_out/bench.log python bench.py --steps 20 --warmup 3 --no-cpu-baseline; tail -2 gpurun_out/bench.log ;;
    cmain)  stage 300 gpurun_out/cmain.log ./build/nbody_main -n 262144 -s 20; tail -5 gpurun_out/cmain.log ;;
    prof)   cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
            rm -rf gpurun_out/prof
            stage 600 gpurun_out/rocprof_stats.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-live-pmc
            tail -3 gpurun_out/rocprof_stats.log; find gpurun_out/prof -name '*stats*' | head ;;
    prof64) cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
            rm -rf gpurun_out/prof64
            stage 600 gpurun_out/rocprof_stats64.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof64 -- python3 bench.py --precision fp64 --steps 10 --warmup 2 --no-cpu-baseline --no-live-pmc
            tail -3 gpurun_out/rocprof_stats64.log; find gpurun_out/prof64 -name '*stats*' | head ;;
    bench64) stage 300 gpurun_out/bench64.log python bench.py --precision fp64 --steps 20 --warmup 3 --no-cpu-baseline; tail -1 gpurun_out/bench64.log ;;
    pmc)    # PMC_ARGS = extra bench.py arguments (another N, --general-mass ...), PMC_DIR = where the passes go (default gpurun_out/pmc)
            cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
            pmcdir=${PMC_DIR:-gpurun_out/pmc}
            rm -rf $pmcdir
            for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY" \
                       "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS" \
                       "GRBM_GUI_ACTIVE GRBM_COUNT FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
              tag=$(echo $set | tr ' ' '_' | cut -c1-40)
              stage 300 gpurun_out/pmc_$tag.log rocprofv3 --pmc $set --output-format csv -d $pmcdir/$tag -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --no-sustained --no-secondary --no-live-pmc ${PMC_ARGS:-}
            done
            find $pmcdir -name '*counter_collection.csv' | head ;;
    *) echo "unknown stage $what" ;;
  esac
done
