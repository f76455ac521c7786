#!/bin/bash
# A/B: material records staged in LDS against read from the global copy (RT_STAGE_MATERIALS), a scene that has room for both.
for s in 1 0 1 0; do
  echo "== RT_STAGE_MATERIALS=$s"
  RT_STAGE_MATERIALS=$s RT_WF_DEBUG=1 timeout -k 10 100 python scripts/gpu_prof_case.py ${1:-clutter_250} 3840 2160 8 u8 2>&1 | grep -E '^\[scene|^clutter' || exit 1
done
