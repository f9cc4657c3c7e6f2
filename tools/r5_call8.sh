#!/bin/bash
mkdir -p gpurun_out/r5c8; O=$(pwd)/gpurun_out/r5c8; export PYTHONUNBUFFERED=1; REPO=$(pwd)
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/st_train && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_train -o run -- python $REPO/tools/train_timing.py 256 5 > $O/stats_train.log 2>&1; find /tmp/st_train -name "*kernel_stats.csv" -exec cp {} $O/train_kernel_stats.csv \; )
head -22 $O/train_kernel_stats.csv | cut -c1-150
