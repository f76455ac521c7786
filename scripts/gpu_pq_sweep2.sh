#!/bin/bash
set -o pipefail
for s in "final 1920 1080 3 spp64" "final_textured_lens 1920 1080 3 spp64" "lens_sampling 1920 1080 3 spp64" "indirect_light 1920 1080 3 spp64"; do
  echo "#### $s"
  RT_WF_DEBUG=1 timeout -k 10 100 python scripts/gpu_prof_case.py $s 2>&1 | grep "^\[scene" | head -1
  bash scripts/gpu_ab.sh "$s" prod pq16 || exit 1
done
