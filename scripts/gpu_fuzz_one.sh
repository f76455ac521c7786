#!/bin/bash
# one fuzz case through the three path-trace kernels: bash scripts/gpu_fuzz_one.sh <seed>
for env in "X=1" "RT_PT_QUEUES=0" "RT_PT_GENERIC=1" "RT_PT_EXACT=1" "RT_PT_CLASSES=0"; do
  echo "== $env"
  env $env FUZZ_ONE=1 timeout -k 10 100 python -u scripts/gpu_fuzz.py 0 $1 2>&1 | grep -v amdgpu
done
