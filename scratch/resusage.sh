#!/bin/bash
# usage: scratch/resusage.sh file.hip  -> one line per kernel: name vgpr agpr scratch occupancy
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c "$1" -o /tmp/ru.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
cur=None
for line in sys.stdin:
    m=re.search(r'remark: (?:\s*)(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)',line)
    if not m: continue
    k,v=m.groups()
    if k=='Function Name':
        if cur: print(cur)
        cur=v[:70]
    else: cur+=f' {k.split()[0]}={v}'
if cur: print(cur)
"
