#!/bin/bash
# VGPR / scratch / LDS usage of the kernels of libraytracer_amd.so (from the code objects' metadata), one
# translation unit at a time: scripts/kernel_regs.sh [object files...]  (default: the production objects)
set -e
L=/opt/rocm/lib/llvm/bin
cd "$(dirname "$0")/../raytracer_amd/csrc"
objs=${@:-render_hip.o k_blockwave.o k_fused.o k_pathwave.o}
for o in $objs; do
  $L/llvm-objcopy --dump-section .hip_fatbin=/tmp/rt_fat.bin "$o" /tmp/rt_unused.o
  $L/clang-offload-bundler --unbundle --type=o --input=/tmp/rt_fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=/tmp/rt_dev.o
  $L/llvm-readelf --notes /tmp/rt_dev.o | python3 -c "
import sys,re
rec={}
for line in sys.stdin:
    m=re.search(r'\.(name|vgpr_count|agpr_count|sgpr_count|private_segment_fixed_size|vgpr_spill_count|group_segment_fixed_size):\s+(\S+)', line)
    if not m: continue
    k,v=m.groups()
    if k=='name' and not v.startswith('_Z') : continue
    rec.setdefault('cur',{})[k]=v
    if len(rec['cur'])==7:
        c=rec.pop('cur'); print(c['name'][:70].ljust(70), 'vgpr',c['vgpr_count'],'agpr',c['agpr_count'],'spill',c['vgpr_spill_count'],'scratch',c['private_segment_fixed_size'],'lds',c['group_segment_fixed_size'])
"
done
