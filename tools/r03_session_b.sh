#!/bin/bash
# Round-3 GPU visit B: full GPU tests (subset draw, new collision kernel, slabs), collision counters of the new kernel,
# fast-mode PMC passes without the small warm-up forward, bench line incl. the whole-config-4 extra.
mkdir -p gpurun_out/r03b
export PYTHONUNBUFFERED=1
REPO=$(pwd); O=$REPO/gpurun_out/r03b
timeout 1800 python -m pytest tests -m gpu -q -rA --tb=short -p no:cacheprovider > $O/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> $O/pytest_gpu.log; grep -E "FAILED|ERROR" $O/pytest_gpu.log | head -20; tail -3 $O/pytest_gpu.log | cut -c1-300
cp gpurun_out/soak_hashes.json gpurun_out/horizon_report.json $O/ 2>/dev/null
pmc() {  # tag, counters, script args
  tag=$1; ctr=$2; shift 2
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc_$tag && timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$tag -o run -- python $REPO/"$@" > $O/pmc_${tag}.log 2>&1
    echo "pmc $tag exit: $?"
    for f in $(find /tmp/pmc_$tag -name "*counter_collection.csv"); do python $REPO/tools/pmc_summary.py $f > $O/pmc_${tag}.csv; done )
}
stats() {  # tag, script args
  tag=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/st_$tag && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$tag -o run -- python $REPO/"$@" > $O/stats_${tag}.log 2>&1
    echo "stats $tag exit: $?"
    find /tmp/st_$tag -name "*kernel_stats.csv" -exec cp {} $O/stats_${tag}_kernel_stats.csv \; )
}
python tools/collision_timing.py 8192 50 20 > $O/collision_c4.json 2> $O/collision_c4.err; cat $O/collision_c4.json
python tools/collision_timing.py 1024 1 50 > $O/collision_c2.json 2> $O/collision_c2.err; cat $O/collision_c2.json
python tools/collision_timing.py 8192 1 50 > $O/collision_step.json 2> $O/collision_step.err; cat $O/collision_step.json
pmc col1 "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" tools/collision_timing.py 8192 50 3
pmc col2 "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32" tools/collision_timing.py 8192 50 3
pmc col3 "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM" tools/collision_timing.py 8192 50 3
stats col tools/collision_timing.py 8192 50 5
timeout 1200 python bench.py --whole-batch-steps 1 > $O/bench.log 2> $O/bench.err; echo "bench exit: $?" >> $O/bench.err; tail -2 $O/bench.err; cut -c1-300 $O/bench.log
