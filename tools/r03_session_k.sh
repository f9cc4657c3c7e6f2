#!/bin/bash
mkdir -p gpurun_out/r03k
B="bench.py --steps 2 --warmup 1 --cpu-envs 0 --extra 0 --fast-steps 0 --pipeline-steps 0 --all-slots-steps 0"
bash tools/pmc_pass.sh sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" $B > /dev/null
bash tools/pmc_pass.sh sq3 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES" $B > /dev/null
bash tools/pmc_pass.sh sq4 "GRBM_GUI_ACTIVE SQ_LEVEL_WAVES SQ_CYCLES SQ_BUSY_CU_CYCLES" $B > /dev/null
for t in sq1 sq3 sq4; do cp gpurun_out/pmc_$t.csv gpurun_out/r03k/; tail -1 gpurun_out/pmc_$t.log | cut -c1-100; done
