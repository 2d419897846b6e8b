#!/bin/bash
# register / scratch / LDS use of the kernels whose name matches $1 (default iterate_col), from hipcc's resource-usage remarks
cd "$(dirname "$0")/../openfx-opencv_amd" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -I../include -c csrc/farneback.hip -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 |
python3 -c '
import re,sys,subprocess
pat=sys.argv[1]
cur=None
for l in sys.stdin:
    m=re.search(r"Function Name: (\S+)",l)
    if m:
        cur=subprocess.run(["c++filt",m.group(1)],capture_output=True,text=True).stdout.strip().split("(float")[0].replace("(anonymous namespace)::",""); vals={}
        continue
    m=re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)",l)
    if m and cur:
        vals[m.group(1).strip()]=int(m.group(2))
        if m.group(1).startswith("LDS Size"):
            if pat in cur: print("%-70s VGPR %3d AGPR %3d scratch %4d spillV %3d spillS %3d occ %d LDS %6d"%(cur[-70:],vals.get("VGPRs",0),vals.get("AGPRs",0),vals.get("ScratchSize [bytes/lane]",0),vals.get("VGPRs Spill",0),vals.get("SGPRs Spill",0),vals.get("Occupancy [waves/SIMD]",0),vals.get("LDS Size [bytes/block]",0)))
            cur=None
' "${1:-iterate_col}"
