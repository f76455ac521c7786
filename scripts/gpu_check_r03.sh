#!/bin/bash
# Full GPU suite, then the default bench run; the bench line is summarised on stdout.  Usage (from the repo root, on the GPU box):
#   bash scripts/gpu_check_r03.sh <tag>
set -o pipefail
tag=${1:-check}
mkdir -p gpurun_out/r03
timeout -k 10 600 python -u -m pytest tests -m gpu -q --timeout 120 --timeout-method thread 2>&1 | tail -4 || exit 1
timeout -k 10 500 python bench.py > gpurun_out/r03/bench_$tag.json 2> gpurun_out/r03/bench_$tag.err || { tail -c 800 gpurun_out/r03/bench_$tag.err; exit 1; }
python scripts/bench_summary.py gpurun_out/r03/bench_$tag.json
