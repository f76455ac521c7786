#!/bin/bash
# Whitted check on the GPU box (run through gpurun from the repo root):
#   scripts/gpu_w_check.sh <tag> [variant library ...]  -> gpurun_out/r02/{wtests,wbench}_<tag>*.{log,json}
# the Whitted parity tests, then the clutter_500 4K bench with the production library and with each variant named.
set -o pipefail
TAG=${1:-x}; shift
mkdir -p gpurun_out/r02
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_rays.py -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/r02/wtests_$TAG.log 2>&1
rc=$?
tail -4 gpurun_out/r02/wtests_$TAG.log
[ $rc -eq 0 ] || exit $rc
for lib in "" "$@"; do
  name=${lib:+_$(basename "$lib" .so)}
  RT_AMD_LIB=${lib:+$PWD/$lib} timeout -k 10 200 python3 bench.py --workload clutter_500_4k --no-cpu-baseline > gpurun_out/r02/wbench_$TAG$name.json 2> gpurun_out/r02/wbench_$TAG$name.err || { tail -5 gpurun_out/r02/wbench_$TAG$name.err; exit 1; }
  python3 -c "
import json
d=json.load(open('gpurun_out/r02/wbench_$TAG$name.json'))
r=d['roofline']
print('${lib:-production}: Mrays/s %.0f  ms/step %.4f  frac %.4f  single-frame ms %.4f' % (d['value'], d['ms_per_step'], r['frac'], r['latency_ms_single_frame']))
"
done
