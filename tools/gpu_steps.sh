#!/bin/bash
# Runs the steps of one gpurun call in sequence, each under its own `timeout -k 10`, logging to gpurun_out/<round>/<name>.log.
# A step that FAILS (assertion, non-zero exit) does not stop the sequence; a step that is KILLED or TIMES OUT (124 / 137 / 143)
# does: no further GPU step is started after a hang.
#   tools/gpu_steps.sh r04 "name1|seconds|command ..." "name2|seconds|command ..."
round=$1; shift
mkdir -p gpurun_out/$round
rc_all=0
for spec in "$@"; do
  name=${spec%%|*}; rest=${spec#*|}; secs=${rest%%|*}; cmd=${rest#*|}
  echo "[gpu_steps] $name (limit ${secs}s): $cmd"
  t0=$(date +%s)
  timeout -k 10 "$secs" bash -o pipefail -c "$cmd" > gpurun_out/$round/$name.log 2>&1
  rc=$?
  echo "[gpu_steps] $name rc=$rc in $(( $(date +%s) - t0 ))s; tail:"
  tail -n 6 gpurun_out/$round/$name.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 143 ]; then
    echo "[gpu_steps] $name was killed / timed out: stopping here"
    exit $rc
  fi
  [ $rc -ne 0 ] && rc_all=$rc
done
exit $rc_all
