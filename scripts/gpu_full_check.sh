#!/bin/bash
# Full GPU check (run through gpurun from the repo root): scripts/gpu_full_check.sh <tag>
#   the whole -m gpu suite, then the default bench line without the CPU leg -> gpurun_out/r02/{suite,bench}_<tag>.*
set -o pipefail
TAG=${1:-x}
mkdir -p gpurun_out/r02
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/r02/suite_$TAG.log 2>&1
rc=$?
tail -4 gpurun_out/r02/suite_$TAG.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 bench.py --no-cpu-baseline > gpurun_out/r02/bench_$TAG.json 2> gpurun_out/r02/bench_$TAG.err || { tail -5 gpurun_out/r02/bench_$TAG.err; exit 1; }
python3 -c "
import json
d=json.load(open('gpurun_out/r02/bench_$TAG.json'))
print('whitted  Mrays/s %.0f  ms/step %.4f  frac %.4f  single-frame ms %s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('latency_ms_single_frame')))
p=d['pathtrace']
print('pathtrace spp/s %.1f  frac %.4f' % (p['value'], p['roofline']['frac']))
"
