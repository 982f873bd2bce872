#!/usr/bin/env bash
# Run a list of GPU steps in order on the gpurun box; each step is bounded by its own timeout
# and logs to gpurun_out/<name>.log.  A step that is killed at its limit (124/137) ends the
# whole list (never start another GPU step after a hang).  Usage:
#   scripts/gpu_steps.sh name1 secs1 'cmd1' name2 secs2 'cmd2' ...
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rc_all=0
while [ $# -ge 3 ]; do
  name=$1; secs=$2; cmd=$3; shift 3
  echo "=== $name (limit ${secs}s): $cmd"
  start=$(date +%s)
  timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "=== $name exit $rc after $(( $(date +%s) - start ))s"
  tail -n 12 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "=== $name hit its limit: stopping"; exit $rc; fi
  [ $rc -ne 0 ] && rc_all=$rc
done
exit $rc_all
