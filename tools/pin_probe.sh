#!/bin/bash
# tools/pin_probe.sh — runs every scenario of tools/pin_probe.hip ONCE, each in its own process, and records the exit
# status (a signal shows as 128 + signo: 134 = SIGABRT).  Output -> gpurun_out/pin_probe.log (copy to profiles/).
set -u
mkdir -p gpurun_out
log=${PIN_LOG:-gpurun_out/pin_probe.log}
{
  echo "pin_probe: $(date -u +%FT%TZ) $(uname -r)"
  /opt/rocm/bin/hipconfig --version 2>/dev/null | head -1
} > $log
for sc in ${PIN_SCENARIOS:-control adjacent adjacent_heap before d2h_adjacent overlap_reg stale_reuse stale_reuse_mmap unreg_freed}; do
  echo "--- $sc" >> $log
  AMD_LOG_LEVEL=1 timeout -k 5 60 ./build/pin_probe $sc >> $log 2>&1
  echo "exit status of $sc: $?" >> $log
done
cat $log
