#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy: bash tools/gpu.sh <timeout-seconds> '<command>'
t=$1; shift
for i in $(seq 1 40); do
  out=$(/usr/local/graft/bin/gpurun --timeout $t -- "$@" 2>&1)
  echo "$out" | tail -100
  if echo "$out" | grep -q "status=transient"; then sleep 60; continue; fi
  break
done
