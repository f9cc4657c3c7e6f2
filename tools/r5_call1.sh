#!/bin/bash
# round 5, GPU call 1: parity gate on the new tests + A/B of the bf16x3 SA2 boundary path
mkdir -p gpurun_out/r5c1; O=gpurun_out/r5c1; export PYTHONUNBUFFERED=1
REPO=$(pwd)
for v in r4sa2 new r4sa2 new; do
  if [ $v = new ]; then L=""; else L=$REPO/build_ab/libmpinets_hip_$v.so; fi
  MPX_LIB_PATH=$L timeout 300 python tools/fast_timing.py 8192 4 > $O/fast_$v.$RANDOM.log 2>&1
done
grep -h 'envs bf16x3\|lib ' $O/fast_*.log
timeout 2400 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider --durations=15 > $O/pytest_gpu.log 2>&1
echo "pytest exit: $?"; tail -25 $O/pytest_gpu.log | cut -c1-220
