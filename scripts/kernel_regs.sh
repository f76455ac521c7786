#!/bin/bash
# VGPR / scratch / LDS usage of the kernels in libraytracer_amd.so (from the code object's metadata)
set -e
lib=${1:-/root/repo/raytracer_amd/csrc/libraytracer_amd.so}
L=/opt/rocm/lib/llvm/bin
$L/llvm-objcopy --dump-section .hip_fatbin=/tmp/rt_fat.bin "$lib"
$L/clang-offload-bundler --unbundle --type=o --input=/tmp/rt_fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=/tmp/rt_dev.o
$L/llvm-readelf --notes /tmp/rt_dev.o | python3 -c "
import sys,re
name=None; rec={}
for line in sys.stdin:
    m=re.search(r'\.(name|vgpr_count|agpr_count|sgpr_count|private_segment_fixed_size|vgpr_spill_count|group_segment_fixed_size):\s+(\S+)', line)
    if not m: continue
    k,v=m.groups()
    if k=='name' and not v.startswith('_Z') : continue
    rec.setdefault('cur',{})[k]=v
    if len(rec['cur'])==7:
        c=rec.pop('cur'); print(c['name'][:70].ljust(70), 'vgpr',c['vgpr_count'],'agpr',c['agpr_count'],'spill',c['vgpr_spill_count'],'scratch',c['private_segment_fixed_size'],'lds',c['group_segment_fixed_size'])
"
