#!/bin/bash
# usage: tools/kres.sh file.hip  -> one line per kernel: name sgpr vgpr agpr scratch lds occupancy
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -pragma-unroll-threshold=1000000 -c "$f" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 \
 | grep -E "Function Name|TotalSGPRs|VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size" \
 | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' \
 | awk '/Function Name/{if(l)print l; l=$0; next}{l=l" | "$0}END{print l}' | c++filt | cut -c1-250
