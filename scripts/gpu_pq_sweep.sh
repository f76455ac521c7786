#!/bin/bash
# pq_kernel: strata per launch, the FIRST frame of a setup (no cost-ordered job table yet) and a later one; pathtrace.json 1080p, 1024 spp
set -o pipefail
for b in 256 512 1024; do
  for n in 1 3; do
    echo "== RT_PQ_BATCH=$b frames $n"
    RT_PQ_BATCH=$b timeout -k 10 200 python scripts/gpu_prof_case.py pathtrace 1920 1080 $n spp1024 || exit 1
  done
done
