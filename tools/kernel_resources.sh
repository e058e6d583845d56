#!/bin/bash
# Register, spill and LDS use of the kernels whose name matches $1 (default k_pair), from hipcc's
# resource remarks.  Extra arguments go to the compiler.
pat=${1:-k_pair}; shift
cd "$(dirname "$0")/../sph_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c sph_amd.hip -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 |
python3 -c "
import sys,re
cur=None
for l in sys.stdin:
    m=re.search(r'remark: +(.*?) \[-Rpass', l)
    if not m: continue
    t=m.group(1)
    if t.startswith('Function Name:'):
        cur=t.split(':',1)[1].strip(); vals={}
    elif cur is not None and ':' in t:
        k,v=t.split(':',1); vals[k.strip()]=v.strip()
        if k.strip().startswith('LDS Size') and '$pat' in cur:
            print(cur[:40], 'VGPR',vals.get('VGPRs'),'SGPR',vals.get('TotalSGPRs'),'scratch',vals.get('ScratchSize [bytes/lane]'),'vspill',vals.get('VGPRs Spill'),'sspill',vals.get('SGPRs Spill'),'occ',vals.get('Occupancy [waves/SIMD]'),'LDS',v.strip())
"
