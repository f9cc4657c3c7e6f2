#!/bin/bash
# the unit claimed late (last tile): parity subset, bench line, the five counter passes of the headline step
mkdir -p gpurun_out/r03lc
export PYTHONUNBUFFERED=1
REPO=$(pwd); O=$REPO/gpurun_out/r03lc
timeout 120 python -m pytest tests/test_gpu_policy.py -m gpu -q --tb=line -p no:cacheprovider -x > $O/pytest_policy.log 2>&1; tail -1 $O/pytest_policy.log
timeout 100 python bench.py --steps 5 --warmup 1 --cpu-envs 0 --extra 0 --fast-steps 0 --pipeline-steps 0 --all-slots-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernels_ms']['sa2_mlp'], d['kernels_ms']['sa1_mlp'])"
PARGS="--envs 8192 --steps 2 --warmup 1 --cpu-envs 0 --extra 0 --fast-steps 0 --pipeline-steps 0 --all-slots-steps 0"
pmc() { tag=$1; ctr=$2; shift 2
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc_$tag && timeout 100 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$tag -o run -- python $REPO/"$@" > $O/pmc_${tag}.log 2>&1
    for f in $(find /tmp/pmc_$tag -name "*counter_collection.csv"); do python $REPO/tools/pmc_summary.py $f > $O/pmc_${tag}.csv; done ) }
pmc head1 "FETCH_SIZE" bench.py $PARGS
grep -E "packed_kernel" $O/pmc_head1.csv | cut -c1-58,95-
pmc head2 "WRITE_SIZE" bench.py $PARGS
pmc head3 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" bench.py $PARGS
pmc head4 "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" bench.py $PARGS
pmc head5 "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" bench.py $PARGS
echo done
