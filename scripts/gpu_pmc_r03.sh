#!/bin/bash
# PMC passes over a few path-traced 1080p frames of pathtrace.json (or another gpu_prof_case.py case), one rocprofv3 run
# per counter set (run through gpurun from the repo root):
#   scripts/gpu_pmc_r03.sh <tag> [case...]   -> gpurun_out/r03/pmc_<tag>.csv
set -e -o pipefail
TAG=${1:-x}
shift || true
CASE=${@:-pathtrace 1920 1080 2}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r03/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d "$OUT/sq" -o pmc -- python3 "$ROOT/scripts/gpu_prof_case.py" $CASE > /dev/null
timeout -s KILL 120 rocprofv3 --output-format csv --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAVES -d "$OUT/sq2" -o pmc -- python3 "$ROOT/scripts/gpu_prof_case.py" $CASE > /dev/null
timeout -s KILL 120 rocprofv3 --output-format csv --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_IFETCH SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR -d "$OUT/sq3" -o pmc -- python3 "$ROOT/scripts/gpu_prof_case.py" $CASE > /dev/null || echo "sq3 pass failed"
timeout -s KILL 120 rocprofv3 --output-format csv --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES -d "$OUT/sq4" -o pmc -- python3 "$ROOT/scripts/gpu_prof_case.py" $CASE > /dev/null || echo "sq4 pass failed"
cd "$ROOT"
python3 scripts/pmc_summary.py "$ROOT/gpurun_out/r03/pmc_$TAG.csv" $(find "$OUT" -name "*counter_collection.csv" | sort)
rm -rf "$OUT"
grep "pt_kernel\|bw_kernel\|pq_kernel\|fw_" "$ROOT/gpurun_out/r03/pmc_$TAG.csv"
