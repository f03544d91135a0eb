#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (exit 3 / "transient"): tools/gpurun_retry.sh TIMEOUT 'command'
T=$1; shift
for i in $(seq 1 40); do
  out=$(/usr/local/graft/bin/gpurun --timeout $T -- "$@" 2>&1); rc=$?
  if echo "$out" | grep -q "status=transient"; then sleep 45; continue; fi
  echo "$out"; exit $rc
done
echo "gave up: GPU slots busy"; exit 3
