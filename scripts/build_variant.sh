#!/bin/bash
# A/B build of the device library with extra compiler flags:
#   scripts/build_variant.sh <name> [flags...]  -> raytracer_amd/csrc/variants/libraytracer_amd_<name>.so
# (load it with RT_AMD_LIB=<path>).  Only the translation units named in TUS are recompiled with the flags
# (default: k_pathwave; constants the host side reads too, like RT_PQ_STATS, need render_hip as well); the others are taken from the production build.
set -e
NAME=$1; shift
cd "$(dirname "$0")/../raytracer_amd/csrc"
make -s -j8 libraytracer_amd.so
mkdir -p variants/$NAME
TUS=${TUS:-k_pathwave}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include"
pids=()
for tu in $TUS; do
  # the production Makefile's per-unit flags, so that a variant differs from production by the flags under test only
  KF=""
  case $tu in k_pathwave|k_pathqueue) KF="-fno-unroll-loops";; esac
  /opt/rocm/bin/hipcc $FLAGS $KF "$@" -c $tu.hip -o variants/$NAME/$tu.o & pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
OBJS=""
for tu in render_hip k_blockwave k_fused k_pathwave k_pathqueue; do
  if [ -f variants/$NAME/$tu.o ]; then OBJS="$OBJS variants/$NAME/$tu.o"; else OBJS="$OBJS $tu.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o variants/libraytracer_amd_$NAME.so $OBJS rt_group.o host_scene.o fbvh_build.o
echo variants/libraytracer_amd_$NAME.so
