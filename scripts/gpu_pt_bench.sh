#!/bin/bash
# one line per run of the path-trace bench: scripts/gpu_pt_bench.sh [bench args...]
timeout -k 10 200 python3 bench.py --workload pathtrace_1080p --spp ${SPP:-64} --pt-steps 3 --pt-warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('spp/s %.1f ms/step %.2f frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))"
