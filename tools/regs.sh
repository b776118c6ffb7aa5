#!/bin/bash
# register / spill report of one kernel source (developer aid): tools/regs.sh cell2x.hip
cd /root/repo/vp-suite_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-gpu-rdc \
  -Rpass-analysis=kernel-resource-usage -save-temps=obj -c "$1" -o /tmp/${1%.hip}.o 2>&1 | \
  grep -E "error|Function Name|TotalSGPRs| VGPRs:|ScratchSize|VGPRs Spill|SGPRs Spill" | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//; s/[a-z0-9_]*.hip:[0-9]*:[0-9]*: remark: //' | paste - - - - - - | sed 's/Function Name: //; s/  */ /g'
