#!/bin/bash
# One GPU-box visit: parity tests (no -x: collect everything), smoke, timings.  Logs -> gpurun_out/
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
( rocm-smi --showproductname 2>/dev/null | head -8; nproc ) > gpurun_out/box.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q -rA --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke exit: $?" >> gpurun_out/smoke.log
timeout 600 python tools/quick_timing.py ${1:-512} > gpurun_out/timing.log 2>&1
echo "timing exit: $?" >> gpurun_out/timing.log
tail -5 gpurun_out/pytest_gpu.log; tail -3 gpurun_out/smoke.log; tail -15 gpurun_out/timing.log
