#!/bin/bash
# Register / scratch / occupancy table of the kernels of one translation unit (compiler remarks, no GPU needed):
#   bash profiles/tools/kernel_regs.sh keypoint-diffusion_amd/csrc/sgemm.hip [name filter] [extra hipcc flags]
src=$1; filt=${2:-.}; shift; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage "$@" -c "$src" -o /dev/null 2>&1 | python3 -c "
import sys, re
cur = None; rows = []
for line in sys.stdin:
    m = re.search(r'remark: (?:[^:]*:)*\s*([A-Za-z ]+?)(?: \[[^]]*\])?: (\S+)', line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2)
    if k == 'Function Name': cur = {'name': v}; rows.append(cur)
    elif cur is not None: cur[k] = v
for r in rows:
    print('%-70s vgpr %4s agpr %4s scratch %5s occ %2s vspill %4s lds %6s' % (r['name'][-70:], r.get('VGPRs'), r.get('AGPRs'), r.get('ScratchSize'), r.get('Occupancy'), r.get('VGPRs Spill'), r.get('LDS Size')))
" | grep -E "$filt"
