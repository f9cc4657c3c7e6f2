#!/bin/bash
mkdir -p gpurun_out/r5c17; O=gpurun_out/r5c17; export PYTHONUNBUFFERED=1; REPO=$(pwd)
timeout 900 python -m pytest tests/test_gpu_policy.py tests/test_gpu_model_golden.py tests/test_gpu_horizon.py -m gpu -q -x --tb=short -p no:cacheprovider -k "bf16x3 and not 50_step" > $O/pytest_bf16.log 2>&1
echo "pytest exit: $?"; tail -4 $O/pytest_bf16.log | cut -c1-220
for v in denseold new denseold new; do
  if [ $v = new ]; then L=""; else L=$REPO/build_ab/libmpinets_hip_$v.so; fi
  MPX_LIB_PATH=$L timeout 300 python tools/fast_timing.py 8192 4 noref > $O/fast_$v.$RANDOM.log 2>&1
done
grep -H 'envs bf16x3' $O/fast_*.log | cut -c30-330
