#!/bin/bash
# Runs a list of GPU steps on the box, each under its own timeout, logging to gpurun_out/<tag>/.  A step that fails goes on to the
# next one; a step that is KILLED at its limit (rc 124 / 137) ends the call: nothing else is started on a GPU that may be wedged.
# usage: tools/gpu_steps.sh <tag> <<'STEPS'
#   name|seconds|command ...
# STEPS
tag=$1
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
while IFS='|' read -r name secs cmd; do
  [ -z "$name" ] && continue
  case "$name" in \#*) continue;; esac
  echo "=== $name (limit ${secs}s): $cmd" | tee -a "$out/steps.log"
  start=$(date +%s)
  # heartbeat: a long step behind a pipe (| tail) writes nothing until it ends, and a call that is silent for minutes is taken to be hung
  ( while sleep 60; do echo "    ... $name running $(( $(date +%s) - start )) s"; done ) &
  hb=$!
  timeout -k 10 "$secs" bash -c "$cmd" > "$out/$name.out" 2> "$out/$name.err" < /dev/null
  rc=$?
  kill $hb 2>/dev/null; wait $hb 2>/dev/null
  echo "    rc $rc after $(( $(date +%s) - start )) s" | tee -a "$out/steps.log"
  tail -n 3 "$out/$name.out" | cut -c1-300
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "    KILLED at its limit: stopping here" | tee -a "$out/steps.log"; exit 3; fi
done
exit 0
