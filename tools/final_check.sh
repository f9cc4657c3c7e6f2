#!/bin/bash
# final check of the tree: GPU suite, smoke, training-step timings + kernel stats (profiles/r06_train_step*)
mkdir -p gpurun_out/r06ev; O=$(pwd)/gpurun_out/r06ev; export PYTHONUNBUFFERED=1; REPO=$(pwd)
timeout 1800 python -m pytest tests -m gpu -q -rA --tb=short -p no:cacheprovider > $O/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
stats() { tag=$1; shift; ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/st_$tag && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$tag -o run -- python $REPO/"$@" > $O/stats_${tag}.log 2>&1; find /tmp/st_$tag -name "*kernel_stats.csv" -exec cp {} $O/stats_${tag}_kernel_stats.csv \; ); }
stats train tools/train_timing.py 256 5
stats trainx3 tools/train_timing.py 256 5 bf16x3
python tools/train_timing.py 256 5 > $O/train_256.log 2>&1; python tools/train_timing.py 10 10 > $O/train_10.log 2>&1
python tools/train_timing.py 256 5 bf16x3 >> $O/train_256.log 2>&1
grep -h 'B=' $O/train_10.log $O/train_256.log
