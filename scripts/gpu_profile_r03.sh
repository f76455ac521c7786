#!/bin/bash
# Round-3 profile set on the GPU box (run through gpurun from the repo root): scripts/gpu_profile_r03.sh <tag>
#   -> gpurun_out/prof_<tag>/...   kernel statistics of the default bench (both workloads) and of clutter_500 with one frame
#   in flight, then PMC passes (one rocprofv3 run each: it does not split counters over passes) on plain frames of
#   clutter_500 at 4K and of pathtrace.json at 1080p (1024 strata x 16 light samples: one launch of pq_kernel).
set -e -o pipefail
TAG=${1:-r03}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ks" -o ks -- python3 "$ROOT/bench.py" --no-cpu-baseline > "$OUT/bench_under_rocprof.json"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ks1" -o ks -- python3 "$ROOT/bench.py" --no-cpu-baseline --workload clutter_500_4k --in-flight 1 > "$OUT/bench_in_flight1_under_rocprof.json"
SQ1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
SQ2="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES GRBM_GUI_ACTIVE"
for case in "clutter_500 3840 2160 3 u8" "pathtrace 1920 1080 2 spp1024"; do
  name=${case%% *}
  timeout -s KILL 180 rocprofv3 --output-format csv --pmc $SQ1 -d "$OUT/pmc_${name}_sq1" -o pmc -- python3 "$ROOT/scripts/gpu_prof_case.py" $case > /dev/null
  timeout -s KILL 180 rocprofv3 --output-format csv --pmc $SQ2 -d "$OUT/pmc_${name}_sq2" -o pmc -- python3 "$ROOT/scripts/gpu_prof_case.py" $case > /dev/null
  timeout -s KILL 180 rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/pmc_${name}_fetch" -o pmc -- python3 "$ROOT/scripts/gpu_prof_case.py" $case > /dev/null
  timeout -s KILL 180 rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$OUT/pmc_${name}_write" -o pmc -- python3 "$ROOT/scripts/gpu_prof_case.py" $case > /dev/null
  cd "$ROOT"
  python3 scripts/pmc_summary.py "$OUT/pmc_${name}.csv" $(find "$OUT"/pmc_${name}_* -name "*counter_collection.csv" | sort)
  cd /tmp
done
cd "$ROOT"
for d in ks ks1; do cp "$(find "$OUT/$d" -name "*kernel_stats.csv" | head -1)" "$OUT/${d}_kernel_stats.csv"; done
rm -rf "$OUT"/ks "$OUT"/ks1 "$OUT"/pmc_*_sq1 "$OUT"/pmc_*_sq2 "$OUT"/pmc_*_fetch "$OUT"/pmc_*_write
timeout -k 10 400 python3 bench.py > "$OUT/bench.json"
ls -la "$OUT"
