#!/bin/bash
mkdir -p gpurun_out/r5full; O=gpurun_out/r5full; export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=8 > $O/pytest_gpu.log 2>&1
echo "pytest exit: $?"; tail -16 $O/pytest_gpu.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
