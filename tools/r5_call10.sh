#!/bin/bash
mkdir -p gpurun_out/r5c10; O=gpurun_out/r5c10; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_training.py -m gpu -q --tb=short -p no:cacheprovider -k "split_bf16 or in_split" -s > $O/pytest_train.log 2>&1
echo "pytest exit: $?"; grep -h 'relative errors\|largest\|bf16x3 training' $O/pytest_train.log; tail -5 $O/pytest_train.log | cut -c1-220
