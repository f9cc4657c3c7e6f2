#!/bin/bash
# Evidence run of a round (one MI355X): GPU tests, bench line (+ whole configs[4] extra), kernel stats, PMC passes of the
# fp32 headline and of the bf16x3 mode, collision counters.  usage (on the box): bash tools/evidence.sh r05 [notests]
# Everything lands in gpurun_out/<round>ev/; tools/collect.sh <round> copies the summaries into profiles/.
R=${1:-r06}
mkdir -p gpurun_out/${R}ev
export PYTHONUNBUFFERED=1
REPO=$(pwd); O=$REPO/gpurun_out/${R}ev
( rocm-smi --showproductname 2>/dev/null | grep -i "card\|gfx" | head -4; echo "host cores: $(nproc)" ) > $O/box.log 2>&1
if [ "$2" != "notests" ]; then
timeout 1800 python -m pytest tests -m gpu -q -rA --tb=short -p no:cacheprovider > $O/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log | cut -c1-200
fi
cp gpurun_out/soak_hashes.json gpurun_out/horizon_report.json $O/ 2>/dev/null
timeout 1200 python bench.py --whole-batch-steps 1 > $O/bench.log 2> $O/bench.err; echo "bench exit: $?" >> $O/bench.err; tail -1 $O/bench.err; cut -c1-200 $O/bench.log
HEAD_ARGS="--steps 3 --warmup 1 --cpu-envs 0 --extra 0 --fast-steps 0 --all-slots-steps 0 --train-steps 0"
pmc() {  # tag, counters, command...
  tag=$1; ctr=$2; shift 2
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc_$tag && timeout 900 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$tag -o run -- python $REPO/"$@" > $O/pmc_${tag}.log 2>&1
    echo "pmc $tag exit: $?"
    for f in $(find /tmp/pmc_$tag -name "*counter_collection.csv"); do python $REPO/tools/pmc_summary.py $f > $O/pmc_${tag}.csv; done )
}
stats() {  # tag, command...
  tag=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/st_$tag && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$tag -o run -- python $REPO/"$@" > $O/stats_${tag}.log 2>&1
    echo "stats $tag exit: $?"
    find /tmp/st_$tag -name "*kernel_stats.csv" -exec cp {} $O/stats_${tag}_kernel_stats.csv \; )
}
stats head bench.py $HEAD_ARGS
PARGS="--envs 8192 --steps 2 --warmup 1 --cpu-envs 0 --extra 0 --fast-steps 0 --all-slots-steps 0 --train-steps 0"
pmc head1 "FETCH_SIZE" bench.py $PARGS
pmc head2 "WRITE_SIZE" bench.py $PARGS
pmc head3 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" bench.py $PARGS
pmc head4 "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" bench.py $PARGS
pmc head5 "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" bench.py $PARGS
pmc fast1 "FETCH_SIZE" tools/fast_timing.py 8192 2 noref
pmc fast2 "WRITE_SIZE" tools/fast_timing.py 8192 2 noref
pmc fast3 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" tools/fast_timing.py 8192 2 noref
pmc fast4 "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" tools/fast_timing.py 8192 2 noref
pmc fast5 "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS" tools/fast_timing.py 8192 2 noref
stats fast tools/fast_timing.py 8192 3 noref
python tools/collision_timing.py 8192 50 20 > $O/collision_c4.json 2> /dev/null; cat $O/collision_c4.json
python tools/collision_timing.py 1024 1 50 > $O/collision_c2.json 2> /dev/null
pmc col1 "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" tools/collision_timing.py 8192 50 3
pmc col2 "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32" tools/collision_timing.py 8192 50 3
pmc col3 "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM" tools/collision_timing.py 8192 50 3
stats col tools/collision_timing.py 8192 50 5
stats train tools/train_timing.py 256 5
python tools/train_timing.py 256 5 > $O/train_256.log 2>&1; python tools/train_timing.py 10 10 > $O/train_10.log 2>&1
python tools/train_timing.py 256 5 bf16x3 >> $O/train_256.log 2>&1
stats trainx3 tools/train_timing.py 256 5 bf16x3
python tools/probes/sa2_bf16_phase_probe.py > $O/sa2_bf16_phase_probe.log 2>&1
echo done
python tools/sa3_front_timing.py 8192 5 > $O/sa3_front_timing.log 2>&1
( hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_pw tools/probes/mfma_power_probe.hip > /dev/null 2>&1 && /tmp/mfma_pw ) > $O/mfma_power_probe.log 2>&1
echo done2
