#!/bin/bash
# A/B timing of library variants on one case: bash scripts/gpu_ab.sh "<case args of gpu_prof_case.py>" <lib> [<lib> ...]
# ("prod" = the production library).  Each variant renders the case in a process of its own.
set -o pipefail
args=$1; shift
for lib in "$@"; do
  if [ "$lib" = prod ]; then unset RT_AMD_LIB; else export RT_AMD_LIB=$PWD/raytracer_amd/csrc/variants/libraytracer_amd_$lib.so; fi
  echo "== $lib"
  timeout -k 10 200 python scripts/gpu_prof_case.py $args || exit 1
done
