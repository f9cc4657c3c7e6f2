#!/bin/bash
mkdir -p gpurun_out/r5c9; O=gpurun_out/r5c9; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_policy.py -m gpu -q -x --tb=short -p no:cacheprovider -k "training or bf16 or chain or split or linear" -s > $O/pytest_train.log 2>&1
echo "pytest exit: $?"; grep -h 'largest\|bf16x3 training' $O/pytest_train.log; tail -12 $O/pytest_train.log | cut -c1-220
python tools/train_timing.py 256 5 > $O/train_256.log 2>&1; tail -1 $O/train_256.log
python tools/train_timing.py 256 5 bf16x3 > $O/train_256x3.log 2>&1; tail -1 $O/train_256x3.log
python tools/train_timing.py 10 10 bf16x3 > $O/train_10x3.log 2>&1; tail -1 $O/train_10x3.log
