"""Collapse a rocprofv3 counter_collection.csv into per-kernel, per-counter sums and launch counts."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
if not rows:
    sys.exit("empty counter file")
cols = rows[0].keys()
kcol = "Kernel_Name" if "Kernel_Name" in cols else [c for c in cols if "ernel" in c and "ame" in c][0]
ccol = "Counter_Name" if "Counter_Name" in cols else [c for c in cols if "ounter" in c and "ame" in c][0]
vcol = "Counter_Value" if "Counter_Value" in cols else [c for c in cols if "alue" in c][0]
dcol = "Dispatch_Id" if "Dispatch_Id" in cols else None
acc = defaultdict(float)
disp = defaultdict(set)
for r in rows:
    k = r[kcol].split("(")[0][:90]
    acc[(k, r[ccol])] += float(r[vcol])
    if dcol:
        disp[(k, r[ccol])].add(r[dcol])
w = csv.writer(sys.stdout)
w.writerow(["kernel", "counter", "sum", "launches", "per_launch"])
for (k, c), v in sorted(acc.items(), key=lambda kv: -kv[1]):
    n = len(disp[(k, c)]) or 1
    w.writerow([k, c, f"{v:.6g}", n, f"{v / n:.6g}"])
