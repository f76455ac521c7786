#!/bin/bash
# LDS counters of the Whitted frame kernels on clutter_500 at 4K (run through gpurun from the repo root):
#   scripts/gpu_pmc_lds.sh <tag> -> gpurun_out/r02/pmc_lds_<tag>.csv
set -e -o pipefail
TAG=${1:-x}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r02/pmc_lds_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --output-format csv --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU -d "$OUT/a" -o pmc -- python3 "$ROOT/scripts/gpu_prof_case.py" clutter_500 3840 2160 3 u8 > /dev/null
timeout -s KILL 120 rocprofv3 --output-format csv --pmc SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU -d "$OUT/b" -o pmc -- python3 "$ROOT/scripts/gpu_prof_case.py" clutter_500 3840 2160 3 u8 > /dev/null
cd "$ROOT"
python3 scripts/pmc_summary.py "$ROOT/gpurun_out/r02/pmc_lds_$TAG.csv" $(find "$OUT" -name "*counter_collection.csv" | sort)
rm -rf "$OUT"
grep "fw_" "$ROOT/gpurun_out/r02/pmc_lds_$TAG.csv"
