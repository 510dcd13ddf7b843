#!/bin/bash
# register / LDS / spill report of every kernel in one HIP source (cross-compile, no GPU).  usage: tools/kres.sh tatt_amd/csrc/gru.hip [filter]
src=$1; filt=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Rpass-analysis=kernel-resource-usage -c $src -o /tmp/kres_$$.o 2>&1 \
 | grep -E "Function Name|  VGPRs:|AGPRs:|ScratchSize|Occupancy|LDS Size" | sed -E 's/.*remark: [^ ]+ +//; s/ \[-Rpass.*//' | paste - - - - - - | grep -E "$filt"
rm -f /tmp/kres_$$.o
