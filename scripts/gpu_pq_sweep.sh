#!/bin/bash
# pq_kernel: strata per launch and waves per workgroup, pathtrace.json 1080p, 256 spp (kernel ms per launch set)
set -o pipefail
for b in 64 128 256; do
  echo "== RT_PQ_BATCH=$b"
  RT_PQ_BATCH=$b timeout -k 10 200 python scripts/gpu_prof_case.py pathtrace 1920 1080 3 spp256 || exit 1
done
for w in 12 16; do
  echo "== RT_PQ_WAVES=$w"
  RT_PQ_WAVES=$w timeout -k 10 200 python scripts/gpu_prof_case.py pathtrace 1920 1080 3 spp256 || exit 1
done
