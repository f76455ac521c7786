#!/bin/bash
# pq_kernel: waves per CU (variant builds pq8 / pq16 against the production 12) and strata per launch, pathtrace.json 1080p, 256 spp
set -o pipefail
bash scripts/gpu_ab.sh 'pathtrace 1920 1080 3 spp256' prod pq8 pq16 || exit 1
for b in 4 16 64 128; do
  echo "== RT_PQ_BATCH=$b"
  RT_PQ_BATCH=$b timeout -k 10 200 python scripts/gpu_prof_case.py pathtrace 1920 1080 3 spp256 || exit 1
done
echo "== RT_PQ_MIN_PIECE=32"
RT_PQ_MIN_PIECE=32 timeout -k 10 200 python scripts/gpu_prof_case.py pathtrace 1920 1080 3 spp256
echo "== RT_PT_QUEUES=0"
RT_PT_QUEUES=0 timeout -k 10 200 python scripts/gpu_prof_case.py pathtrace 1920 1080 3 spp256
