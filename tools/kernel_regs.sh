#!/bin/bash
# tools/kernel_regs.sh file.hip [pattern]: VGPRs, spills, scratch and LDS of each kernel of one source file (gfx950)
f=$1; pat=${2:-.}
cd /tmp && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -x hip -c "$f" -o /tmp/_regs.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|Name:|VGPRs:|VGPR Spill|ScratchSize|Occupancy|LDS Size" | sed -E 's/^[^ ]+ +//; s/ \[-Rpass.*//' \
 | awk '/Name:/ {if (line) print line; line=$0; next} {line=line " | " $0} END {print line}' | grep -E "$pat"
