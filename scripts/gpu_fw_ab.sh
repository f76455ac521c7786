#!/bin/bash
# Whitted A/B of library variants through the bench loop (4 frames in flight) and one frame alone
set -o pipefail
mkdir -p gpurun_out/r03
for lib in "$@"; do
  if [ "$lib" = prod ]; then unset RT_AMD_LIB; else export RT_AMD_LIB=$PWD/raytracer_amd/csrc/variants/libraytracer_amd_$lib.so; fi
  timeout -k 10 300 python bench.py --workload clutter_500_4k --no-cpu-baseline > gpurun_out/r03/bench_fw_$lib.json || exit 1
  python -c "
import json,sys
d=[json.loads(l) for l in open('gpurun_out/r03/bench_fw_$lib.json') if l.startswith('{')][0]
print('$lib', 'ms_per_step %.4f' % d['ms_per_step'], 'single frame %.4f' % d['roofline']['latency_ms_single_frame'])"
done
