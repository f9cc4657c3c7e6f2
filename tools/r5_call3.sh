#!/bin/bash
mkdir -p gpurun_out/r5c5; O=gpurun_out/r5c5; export PYTHONUNBUFFERED=1
REPO=$(pwd)
timeout 900 python -m pytest tests/test_gpu_policy.py tests/test_gpu_model_golden.py tests/test_gpu_edge_cases.py -m gpu -q -x --tb=short -p no:cacheprovider -k "bf16x3 or elision or hits_only or small_batch" > $O/pytest_bf16.log 2>&1
echo "pytest exit: $?"; tail -5 $O/pytest_bf16.log | cut -c1-220
python tools/probes/sa2_bf16_phase_probe.py > $O/probe_new.log 2>&1; tail -6 $O/probe_new.log | cut -c1-300
for v in defer0 new defer0 new; do
  if [ $v = new ]; then L=""; else L=$REPO/build_ab/libmpinets_hip_$v.so; fi
  MPX_LIB_PATH=$L timeout 300 python tools/fast_timing.py 8192 4 noref > $O/fast_$v.$RANDOM.log 2>&1
done
grep -H 'envs bf16x3' $O/fast_*.log | cut -c1-160
