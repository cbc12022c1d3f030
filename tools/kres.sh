#!/bin/bash
# Per-kernel register / LDS / occupancy table of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage), CPU only.
# usage: tools/kres.sh mscnn_amd/csrc/conv.hip [filter-regex]
f=$1; pat=${2:-.}
cd "$(dirname "$f")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -D__HIP_PLATFORM_AMD__ $EXTRA \
  -Rpass-analysis=kernel-resource-usage -c "$(basename "$f")" -o /tmp/kres_$$.o 2>&1 \
 | grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|Occupancy|LDS Size" \
 | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' \
 | awk '/Function Name/{if(n)print n,v,a,s,o,l; n=$3} /^ *VGPRs:/{v="vgpr="$2} /AGPRs:/{a="agpr="$2} /ScratchSize/{s="scratch="$3} /Occupancy/{o="occ="$3} /LDS Size/{l="lds="$4} END{print n,v,a,s,o,l}' \
 | c++filt | grep -E "$pat" | sed -e 's/(anonymous namespace):://g' -e 's/void //'
rm -f /tmp/kres_$$.o
