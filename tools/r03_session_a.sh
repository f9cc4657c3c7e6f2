#!/bin/bash
# Round-3 GPU visit A: GPU tests (incl. soak + RCCL), fast-mode PMC passes at 8192 envs, collision-kernel counters, bench line.
mkdir -p gpurun_out/r03a
export PYTHONUNBUFFERED=1
REPO=$(pwd); O=$REPO/gpurun_out/r03a
( rocm-smi --showproductname 2>/dev/null | grep -i "card\|gfx" | head -4; echo "host cores: $(nproc)" ) > $O/box.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q -rA --tb=short -p no:cacheprovider > $O/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log | cut -c1-300
cp gpurun_out/soak_hashes.json $O/ 2>/dev/null
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
# ---- fast mode (bf16x3) at the bench size
pmc fast1 "FETCH_SIZE" tools/fast_timing.py 8192 2
pmc fast2 "WRITE_SIZE" tools/fast_timing.py 8192 2
pmc fast3 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" tools/fast_timing.py 8192 2
pmc fast4 "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" tools/fast_timing.py 8192 2
pmc fast5 "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS" tools/fast_timing.py 8192 2
stats fast tools/fast_timing.py 8192 3
# ---- collision validation (configs 4 and 2)
python tools/collision_timing.py 8192 50 20 > $O/collision_c4.json 2> $O/collision_c4.err; cat $O/collision_c4.json
python tools/collision_timing.py 1024 1 50 > $O/collision_c2.json 2> $O/collision_c2.err; cat $O/collision_c2.json
pmc col1 "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" tools/collision_timing.py 8192 50 3
pmc col2 "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32" tools/collision_timing.py 8192 50 3
pmc col3 "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM" tools/collision_timing.py 8192 50 3
stats col tools/collision_timing.py 8192 50 5
# ---- the bench line with the new fields
timeout 900 python bench.py > $O/bench.log 2> $O/bench.err; echo "bench exit: $?" >> $O/bench.err; tail -2 $O/bench.err; cut -c1-400 $O/bench.log
