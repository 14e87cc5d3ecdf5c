#!/usr/bin/env bash
# gpurun with a fault budget: three GPU runs that fault inside 15 minutes close gpurun for the whole round (round 3 lost
# 268 GPU-minutes that way).  This wrapper keeps a log of runs that ended in a GPU fault (exit 134 / "Memory access
# fault" in the tail) under gpurun_out/.fault_log and REFUSES to start another run while two faults lie inside the last
# 20 minutes of the wall clock -- it prints how long to wait instead.  Use it for every run that might fault.
#   tools/gpurun_guarded.sh --timeout 600 -- '<command>'
set -u
LOG=gpurun_out/.fault_log
mkdir -p gpurun_out
now=$(date +%s)
recent=0
if [ -f "$LOG" ]; then
  while read -r t; do
    [ -n "$t" ] && [ $((now - t)) -lt 1200 ] && recent=$((recent + 1))
  done < "$LOG"
fi
if [ "$recent" -ge 2 ]; then
  oldest=$(tail -n 2 "$LOG" | head -n 1)
  echo "gpurun_guarded: $recent faulting runs in the last 20 minutes; wait $((1200 - (now - oldest))) s before the next risky run" >&2
  exit 9
fi
out=$(/usr/local/graft/bin/gpurun "$@" 2>&1)
rc=$?
echo "$out"
if echo "$out" | grep -q "memory-access fault\|Memory access fault"; then
  date +%s >> "$LOG"
  echo "gpurun_guarded: fault recorded at $(date -u +%H:%M:%S) UTC ($((recent + 1)) in the last 20 minutes)" >&2
fi
exit $rc
