#!/bin/bash
# gpurun wrapper: retries ONLY while the pool has no free box (status=transient, nothing ran, nothing charged)
T=${GR_TIMEOUT:-900}
for i in 1 2 3 4 5 6 7 8 9 10; do
  out=$(/usr/local/graft/bin/gpurun --timeout $T -- "$@" 2>&1)
  if echo "$out" | grep -q "status=transient"; then sleep 75; continue; fi
  echo "$out"; exit 0
done
echo "$out"
