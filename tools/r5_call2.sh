#!/bin/bash
mkdir -p gpurun_out/r5c2; O=gpurun_out/r5c2; export PYTHONUNBUFFERED=1
REPO=$(pwd)
python tools/probes/sa2_bf16_phase_probe.py > $O/probe_new.log 2>&1; cat $O/probe_new.log | cut -c1-400
MPX_LIB_PATH=$REPO/build_ab/libmpinets_hip_twoacc0.so python tools/probes/sa2_bf16_phase_probe.py > $O/probe_twoacc0.log 2>&1; cat $O/probe_twoacc0.log | cut -c1-400
for v in twoacc0 new; do
  if [ $v = new ]; then L=""; else L=$REPO/build_ab/libmpinets_hip_$v.so; fi
  MPX_LIB_PATH=$L timeout 300 python tools/fast_timing.py 8192 4 noref > $O/fast_$v.log 2>&1
done
grep -h 'envs bf16x3' $O/fast_*.log | cut -c1-120
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --deselect tests/test_gpu_horizon.py::test_50_step_rollout_vs_oracle > $O/pytest_gpu.log 2>&1
echo "pytest exit: $?"; tail -12 $O/pytest_gpu.log | cut -c1-220
