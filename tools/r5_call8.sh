#!/bin/bash
mkdir -p gpurun_out/r5c16; O=$(pwd)/gpurun_out/r5c16; export PYTHONUNBUFFERED=1; REPO=$(pwd)
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/st_train && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_train -o run -- python $REPO/tools/train_timing.py 256 5 bf16x3 > $O/stats_train.log 2>&1; find /tmp/st_train -name "*kernel_stats.csv" -exec cp {} $O/train_x3_kernel_stats.csv \; )
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r5c16/train_x3_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total ms per step (7 steps):', tot/7/1e6)
for r in rows[:16]:
    print('%-70s calls %4s  %.2f ms/step  %.1f%%'%(r['Name'][:70], r['Calls'], float(r['TotalDurationNs'])/7/1e6, float(r['Percentage'])))
PY
