#!/usr/bin/env python3
"""Static map of one kernel's ISA: barriers, loops (backward branches) with their instruction mix and spill traffic.
usage: llvm-objdump -d dev.o > k.s ; scripts/isa_loops.py k.s <substring of the kernel's symbol>"""
import re, sys
src, want = sys.argv[1], sys.argv[2]
lines = open(src).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^[0-9a-f]+ <', l) and want in l)
end = next((i for i in range(start + 1, len(lines)) if re.match(r'^[0-9a-f]+ <', lines[i])), len(lines))
base = int(lines[start].split()[0], 16)
ins = []
for l in lines[start + 1:end]:
    m = re.match(r'\s+(\S+)\s*(.*?)//\s*([0-9A-F]+):', l)
    if not m: continue
    tgt = None
    t = re.search(r'\+0x([0-9a-f]+)>\s*$', l)
    if t: tgt = int(t.group(1), 16)
    ins.append((int(m.group(3), 16) - base, m.group(1), m.group(2).strip(), tgt))
off2k = {o: k for k, (o, _, _, _) in enumerate(ins)}
print(f'{len(ins)} instructions')
bars = [k for k, i in enumerate(ins) if i[1] == 's_barrier']
print('barriers at', bars)
def classes(a, b):
    c = {'valu': 0, 'readlane': 0, 'writelane': 0, 'salu': 0, 'smem': 0, 'mem': 0}
    for _, op, _, _ in ins[a:b]:
        if op.startswith('v_readlane'): c['readlane'] += 1
        elif op.startswith('v_writelane'): c['writelane'] += 1
        elif op.startswith(('s_load', 's_buffer_load')): c['smem'] += 1
        elif op.startswith('s_'): c['salu'] += 1
        elif op.startswith('v_'): c['valu'] += 1
        else: c['mem'] += 1
    return c
edges = [0] + bars + [len(ins)]
for a, b in zip(edges, edges[1:]):
    print(f'between barriers [{a},{b}) n={b - a}', classes(a, b))
if '--phases' in sys.argv: sys.exit(0)
loops = []
for k, (o, op, a, tgt) in enumerate(ins):
    if tgt is not None and op.startswith(('s_cbranch', 's_branch')) and tgt <= o and tgt in off2k:
        loops.append((off2k[tgt], k))
def mix(a, b):
    c = {'n': b - a + 1, 'f64': 0, 'trans': 0, 'imul': 0, 'sld': 0, 'sst': 0, 'vmem': 0, 'lds': 0, 'smem': 0, 'br': 0}
    for _, op, _, _ in ins[a:b + 1]:
        if op.startswith('scratch_load'): c['sld'] += 1
        elif op.startswith('scratch_store'): c['sst'] += 1
        elif op.startswith(('global_', 'flat_', 'buffer_')): c['vmem'] += 1
        elif op.startswith('ds_'): c['lds'] += 1
        elif op.startswith('s_load') or op.startswith('s_buffer_load'): c['smem'] += 1
        elif op.startswith(('s_cbranch', 's_branch')): c['br'] += 1
        if '_f64' in op: c['f64'] += 1
        if op.startswith(('v_rcp', 'v_rsq', 'v_sqrt')): c['trans'] += 1
        if op.startswith(('v_mad_u64', 'v_mul_lo_u32', 'v_mul_hi_u32', 'v_mad_i64')): c['imul'] += 1
    return c
for a, b in sorted(set(loops)):
    depth = sum(1 for x, y in set(loops) if x <= a and b <= y) - 1
    print('  ' * depth + f'loop [{a},{b}] ', mix(a, b))
