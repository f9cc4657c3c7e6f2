#!/bin/bash
# One GPU-box visit.  usage: tools/gpu_session.sh [tests] [smoke] [timing[=B]] [bench[=ARGS]] [prof[=ARGS]]
# Logs -> gpurun_out/ (merged back by gpurun).
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
REPO=$(pwd)
( rocm-smi --showproductname 2>/dev/null | grep -i "card\|gfx" | head -4; echo "host cores: $(nproc)"; free -g | head -2 ) > gpurun_out/box.log 2>&1
for arg in "$@"; do
  key=${arg%%=*}; val=""; [[ "$arg" == *=* ]] && val=${arg#*=}
  case $key in
    tests)
      timeout 1500 python -m pytest tests -m gpu -q -rA --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
      echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log; tail -4 gpurun_out/pytest_gpu.log ;;
    smoke)
      timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
      echo "smoke exit: $?" >> gpurun_out/smoke.log; tail -2 gpurun_out/smoke.log ;;
    timing)
      timeout 900 python tools/quick_timing.py ${val:-512} > gpurun_out/timing.log 2>&1
      echo "timing exit: $?" >> gpurun_out/timing.log; tail -14 gpurun_out/timing.log ;;
    bench)
      timeout 1200 python bench.py $val > gpurun_out/bench.log 2> gpurun_out/bench.err
      echo "bench exit: $?" >> gpurun_out/bench.err; tail -3 gpurun_out/bench.err; cat gpurun_out/bench.log ;;
    prof)
      ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o run -- python $REPO/bench.py ${val:---steps 3 --warmup 1 --cpu-envs 0 --extra 0 --fast-steps 0 --pipeline-steps 0} > $REPO/gpurun_out/prof_bench.log 2> $REPO/gpurun_out/prof.err
        echo "prof exit: $?" >> $REPO/gpurun_out/prof.err
        mkdir -p $REPO/gpurun_out/prof; find /tmp/prof -name "*stats*" -exec cp {} $REPO/gpurun_out/prof/ \; ; ls -la /tmp/prof/* | head -20 >> $REPO/gpurun_out/prof.err )
      tail -3 gpurun_out/prof.err; cat gpurun_out/prof_bench.log | tail -1 | cut -c1-200; head -12 gpurun_out/prof/*kernel_stats.csv ;;
    pmc)
      # one counter group per pass, kernel-trace only (never mixed with other trace domains)
      i=0
      for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY"; do
        i=$((i+1))
        ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc$i && timeout 900 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc$i -o run -- python $REPO/bench.py ${val:---envs 8192 --steps 2 --warmup 1 --cpu-envs 0 --extra 0 --fast-steps 0 --pipeline-steps 0} > $REPO/gpurun_out/pmc${i}_bench.log 2> $REPO/gpurun_out/pmc$i.err
          echo "pmc$i exit: $?" >> $REPO/gpurun_out/pmc$i.err
          mkdir -p $REPO/gpurun_out/pmc; for f in $(find /tmp/pmc$i -name "*counter_collection.csv"); do python $REPO/tools/pmc_summary.py $f > $REPO/gpurun_out/pmc/pass${i}_summary.csv 2>> $REPO/gpurun_out/pmc$i.err; done )
        tail -2 gpurun_out/pmc$i.err; head -20 gpurun_out/pmc/pass${i}_summary.csv
      done ;;
  esac
done
