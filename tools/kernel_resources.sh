#!/bin/bash
# Per-kernel VGPR / SGPR / scratch / LDS of one .hip file (cross-compiles for gfx950, no GPU): worst cases first.
# usage: tools/kernel_resources.sh dxt-lossless-transform_amd/csrc/batch_kernels.hip [grep pattern]
set -e
src=$(readlink -f "$1"); pat=${2:-.}
cd /tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -c "$src" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 |
  python3 -c '
import re,sys,subprocess
rows=[];cur={}
for l in sys.stdin:
    m=re.search(r"remark: +(Function Name|VGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (.*?) \[-Rpass", l)
    if not m: continue
    k,v=m.group(1),m.group(2)
    if k=="Function Name":
        cur={"name":v}; rows.append(cur)
    else: cur[k.split()[0]]=v
names=subprocess.run(["c++filt"],input="\n".join(r["name"] for r in rows),capture_output=True,text=True).stdout.split("\n")
for r,n in zip(rows,names): r["name"]=n
rows.sort(key=lambda r:(-int(r.get("ScratchSize",0)),-int(r.get("VGPRs",0))))
for r in rows: print(r.get("VGPRs"),r.get("TotalSGPRs"),r.get("ScratchSize"),r.get("LDS"),r.get("Occupancy"),r["name"][:150])
' | grep -E "$pat" | head -${3:-25}
