#!/bin/bash
# kernel resource usage summary for kernels matching $1 (regex on the mangled name)
cd /root/repo/pecaller_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-honor-nans -mno-amdgpu-ieee -fPIC -shared -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-variable -o ../libpemap_hip.so pemap_capi.hip pecall_capi.hip -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
pat=re.compile(sys.argv[1])
cur=None
for l in sys.stdin:
    if 'error' in l: print(l.rstrip())
    m=re.search(r'Function Name: (\S+)',l)
    if m:
        cur=m.group(1) if pat.search(m.group(1)) else None
        if cur: print(cur[:44], end=' ')
        continue
    if cur:
        m=re.search(r'remark:\s+(VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)',l)
        if m: print(m.group(1).split()[0][:7]+'='+m.group(2), end=' ')
        if 'LDS Size' in l: print()
" "$1"
