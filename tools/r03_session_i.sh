#!/bin/bash
# where do SA2's waves wait?  two SQ counter passes over the headline step
mkdir -p gpurun_out/r03i
B="bench.py --steps 2 --warmup 1 --cpu-envs 0 --extra 0 --fast-steps 0 --pipeline-steps 0 --all-slots-steps 0"
bash tools/pmc_pass.sh sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" $B > /dev/null
bash tools/pmc_pass.sh sq2 "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_LDS" $B > /dev/null
bash tools/pmc_pass.sh sq3 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH" $B > /dev/null
for t in sq1 sq2 sq3; do cp gpurun_out/pmc_$t.csv gpurun_out/r03i/; tail -2 gpurun_out/pmc_$t.log | cut -c1-200; done
grep -E "packed_kernel|sa3_chain" gpurun_out/r03i/*.csv | cut -d: -f2 | awk -F, '{printf "%-50s %-28s top_mean %s n %s\n", substr($1,1,50), $2, $6, $7}'
