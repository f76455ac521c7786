#!/bin/bash
# one fuzz case at several recursion limits: bash scripts/gpu_fuzz_depths.sh <seed> [<seed> ...]
for seed in "$@"; do
  for d in 2 3 4 5 6 8 10; do
    FUZZ_DEPTH=$d FUZZ_ONE=1 timeout -k 10 100 python -u scripts/gpu_fuzz.py 0 $seed 2>&1 | grep -E "^(ok|FAIL)|pixels off" | tr '\n' ' ' | cut -c1-260; echo
  done
done
