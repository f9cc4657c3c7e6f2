#!/bin/bash
# FETCH_SIZE of the narrow module's kernel with 16 instead of 32 queries per unit (A/B build q16): the L2-window reading of
# DESIGN.md section 5 (environments in flight per XCD) predicts about a third of the fetch
REPO=$(pwd); mkdir -p gpurun_out/l2w
B="bench.py --steps 2 --warmup 1 --cpu-envs 0 --extra 0 --fast-steps 0 --pipeline-steps 0 --all-slots-steps 0"
export PMC_TIMEOUT=120
MPX_LIB_PATH=$REPO/build_ab/libmpinets_hip_q16.so bash tools/pmc_pass.sh q16 "FETCH_SIZE" $B > /dev/null
grep -E "packed_kernel<1" gpurun_out/pmc_q16.csv; cp gpurun_out/pmc_q16.csv gpurun_out/l2w/
