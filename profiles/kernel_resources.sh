#!/bin/bash
# VGPR / SGPR / LDS / scratch per kernel, from the compiler's own resource report for the gfx950 code object
# (hipcc -Rpass-analysis=kernel-resource-usage on the one translation unit): bash profiles/kernel_resources.sh > profiles/r06/kernel_resources.txt
cd "$(dirname "$0")/../josefine_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -c josefine_gpu.hip -o /tmp/jg_dev.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys, re
cur = None; rows = {}
for ln in sys.stdin:
    m = re.search(r'Function Name: (\S+)', ln)
    if m: cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r'remark:\s+([\w \[\]/]+?):\s+(\S+) \[-Rpass', ln)
    if m and cur: rows[cur][m.group(1).strip()] = m.group(2)
import subprocess
def dem(n):
    try: return subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip().replace('(anonymous namespace)::', '').split('(')[0]
    except Exception: return n
print('kernel, VGPRs, AGPRs, SGPRs, scratch B/lane, LDS B/workgroup, occupancy waves/SIMD')
lib = [k for k in rows if 'rocprim' in k]
print('# (%d library kernel instantiations: round 5 removed the last rocPRIM call)' % len(lib))
for k, v in sorted(((k, v) for k, v in rows.items() if 'rocprim' not in k), key=lambda kv: dem(kv[0])):
    print(', '.join([dem(k), v.get('VGPRs','?'), v.get('AGPRs','?'), v.get('TotalSGPRs','?'), v.get('ScratchSize [bytes/lane]','?'), v.get('LDS Size [bytes/block]','?'), v.get('Occupancy [waves/SIMD]','?')]))
"
