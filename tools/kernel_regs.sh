#!/bin/bash
# register / LDS / spill figures of the kernels of one object file: tools/kernel_regs.sh build/conv_igemm.o [name filter]
obj=${1:?object file}; filt=${2:-.}
tmp=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin "$obj" $tmp/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$tmp/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/dev.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $tmp/dev.co | python3 -c "
import sys, re
cur = {}
rows = []
for l in sys.stdin:
    m = re.match(r'\s+-?\s*\.(\w+):\s+(.*)', l)
    if not m: continue
    k, v = m.group(1), m.group(2).strip()
    if k == 'agpr_count' and cur.get('name'): rows.append(cur); cur = {}
    if k in ('name', 'vgpr_count', 'agpr_count', 'sgpr_count', 'vgpr_spill_count', 'sgpr_spill_count', 'group_segment_fixed_size', 'private_segment_fixed_size', 'max_flat_workgroup_size'):
        cur[k] = v
if cur.get('name'): rows.append(cur)
import subprocess
for r in rows:
    n = subprocess.run(['c++filt', r.get('name','')], capture_output=True, text=True).stdout.strip()
    if re.search(sys.argv[1], n):
        print('%-90s vgpr %s agpr %s sgpr %s spill %s lds %s scratch %s wg %s' % (n[:90], r.get('vgpr_count'), r.get('agpr_count'), r.get('sgpr_count'), r.get('vgpr_spill_count'), r.get('group_segment_fixed_size'), r.get('private_segment_fixed_size'), r.get('max_flat_workgroup_size')))
" "$filt"
rm -rf $tmp
