#!/bin/bash
# usage: scratch/kres.sh <file.hip> [extra flags]: compile one source for gfx950 and print VGPRs / scratch per kernel
set -e
SRC=$1; shift
B=$(basename $SRC .hip)
OUT=/tmp/kres_$B; mkdir -p $OUT
( cd /root/repo/textboost_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -munsafe-fp-atomics "$@" -c $B.hip -o $OUT/$B.o -save-temps=obj )
python3 - $OUT/$B-hip-amdgcn-amd-amdhsa-gfx950.s <<'PY'
import re,sys,subprocess
txt=open(sys.argv[1]).read()
ks=re.findall(r'\.amdhsa_kernel (\S+).*?\.amdhsa_next_free_vgpr (\d+).*?\.end_amdhsa_kernel', txt, re.S)
sc=re.findall(r'; ScratchSize: (\d+)', txt)
occ=re.findall(r'; Occupancy: (\d+)', txt)
lds=re.findall(r'; LDSByteSize: (\d+)', txt)
for i,(n,v) in enumerate(ks):
    try: dn=subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt',n],capture_output=True,text=True).stdout.strip()
    except Exception: dn=n
    dn=re.sub(r'\(.*','',dn)[:70]
    print(f"{dn:72s} vgpr {v:>4s} scratch {sc[i] if i<len(sc) else '?':>4s} occ {occ[i] if i<len(occ) else '?'}")
PY
