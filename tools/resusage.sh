#!/bin/bash
# prints VGPR/SGPR/occupancy per kernel of one HIP file: tools_resusage.sh csrc/farneback.hip
cd "$(dirname "$0")/../openfx-opencv_amd"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I../include -c "$1" -o /tmp/_ru.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size" | sed -E 's/.*remark: [^ ]+ +//; s/ \[-Rpass.*//' | paste -d' ' - - - - - - | sed -E 's/Function Name: _ZN12_GLOBAL__N_1[0-9]+//' | cut -c1-200
