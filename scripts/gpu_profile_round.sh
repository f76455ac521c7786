#!/bin/bash
# Collects the round's profile set on the GPU box (run through gpurun from the repo root):
#   scripts/gpu_profile_round.sh <tag>      e.g. r01_g  -> gpurun_out/prof_<tag>/...
# kernel statistics of the default bench (4 frames in flight) and of --in-flight 1 (one launch owns
# the chip: the per-launch durations the roofline is priced on), then PMC passes (one run each:
# rocprofv3 does not split counters over passes) on a few plain frames of clutter_500 at 4K.
set -e -o pipefail
TAG=${1:-round}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ks4" -o ks -- python3 "$ROOT/bench.py" --no-cpu-baseline --steps 20 > "$OUT/bench_in_flight4_under_rocprof.json"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ks1" -o ks -- python3 "$ROOT/bench.py" --no-cpu-baseline --steps 20 --in-flight 1 > "$OUT/bench_in_flight1_under_rocprof.json"
timeout -s KILL 120 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d "$OUT/pmc_sq" -o pmc -- python3 "$ROOT/scripts/gpu_prof_case.py" clutter_500 3840 2160 3 u8 > /dev/null
timeout -s KILL 120 rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o pmc -- python3 "$ROOT/scripts/gpu_prof_case.py" clutter_500 3840 2160 3 u8 > /dev/null
timeout -s KILL 120 rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$OUT/pmc_write" -o pmc -- python3 "$ROOT/scripts/gpu_prof_case.py" clutter_500 3840 2160 3 u8 > /dev/null
cd "$ROOT"
python3 scripts/pmc_summary.py "$OUT/pmc_summary.csv" $(find "$OUT"/pmc_* -name "*counter_collection.csv" | sort)
for d in ks4 ks1; do cp "$(find "$OUT/$d" -name "*kernel_stats.csv" | head -1)" "$OUT/${d}_kernel_stats.csv"; done
rm -rf "$OUT"/ks4 "$OUT"/ks1 "$OUT"/pmc_sq "$OUT"/pmc_fetch "$OUT"/pmc_write
timeout -k 10 200 python3 bench.py --steps 20 > "$OUT/bench.json"
timeout -k 10 200 python3 bench.py --steps 20 --in-flight 1 --no-cpu-baseline > "$OUT/bench_in_flight1.json"
ls -la "$OUT"
