#!/bin/bash
# Path-trace check on the GPU box (run through gpurun from the repo root):
#   scripts/gpu_pt_check.sh <tag> [spp]   -> gpurun_out/r02/{tests,bench}_<tag>.*
# the path-trace parity tests, then the 1080p bench of pathtrace.json with <spp> strata per step.
set -o pipefail
TAG=${1:-x}
SPP=${2:-64}
mkdir -p gpurun_out/r02
timeout -k 10 300 python -u -m pytest tests/test_gpu_pathtrace.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r02/tests_$TAG.log 2>&1
rc=$?
tail -12 gpurun_out/r02/tests_$TAG.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python3 bench.py --workload pathtrace_1080p --spp $SPP --pt-steps 3 --pt-warmup 1 --no-cpu-baseline > gpurun_out/r02/bench_$TAG.json 2> gpurun_out/r02/bench_$TAG.err || { tail -5 gpurun_out/r02/bench_$TAG.err; exit 1; }
python3 -c "
import json,sys
d=json.load(open('gpurun_out/r02/bench_$TAG.json'))
print('spp/s %.1f  ms/step %.2f  Grays/s %.2f  frac %.4f' % (d['value'], d['ms_per_step'], d['mrays_per_s']/1e3, d['roofline']['frac']))
"
