#!/bin/bash
# VGPR count / spills / scratch of every kernel in a HIP source (hipcc -Rpass-analysis=kernel-resource-usage):
#   scripts/kernel_regs.sh stain2stain_amd/csrc/conv3x3_mfma.hip [filter]
# Run after every kernel edit: a spill (scratch > 0) in a hot loop costs more than any restructuring gains.
SRC=$(realpath "$1"); FILT=${2:-.}
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 ${EXTRA_FLAGS} -c "$SRC" -o /tmp/_regs.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|    VGPRs:|VGPRs Spill|ScratchSize|LDS Size" | paste - - - - - \
  | sed -E 's/.*Function Name: ([^ ]*) .*VGPRs: ([0-9]+).*ScratchSize[^:]*: ([0-9]+).*VGPRs Spill: ([0-9]+).*LDS Size[^:]*: ([0-9]+).*/\1 vgpr=\2 scratch=\3 spill=\4 lds=\5/' \
  | grep -E "$FILT" | c++filt | sed -E 's/\(anonymous namespace\):://; s/\(Conv3x3Args\)//; s/\(WgradArgs\)//'
