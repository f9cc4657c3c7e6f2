"""Collapse a rocprofv3 counter_collection.csv into per-kernel, per-counter figures: sum and launch count over all
dispatches, and -- because warm-up launches of a few environments share the kernel name with the full-size ones -- the
mean over the FULL-SIZE dispatches only (`top_mean` over the `top_n` dispatches whose value is >= half of the largest)."""
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
per = defaultdict(lambda: defaultdict(float))  # (kernel, counter) -> dispatch -> value (summed over XCDs / SEs)
for i, r in enumerate(rows):
    k = r[kcol].split("(")[0][:90]
    per[(k, r[ccol])][r[dcol] if dcol else i] += float(r[vcol])
w = csv.writer(sys.stdout)
w.writerow(["kernel", "counter", "sum", "launches", "per_launch", "top_mean", "top_n"])
for (k, c), d in sorted(per.items(), key=lambda kv: -sum(kv[1].values())):
    vals = list(d.values())
    v, n, mx = sum(vals), len(vals), max(vals)
    top = [x for x in vals if x >= 0.5 * mx] if mx > 0 else vals
    w.writerow([k, c, f"{v:.6g}", n, f"{v / n:.6g}", f"{sum(top) / len(top):.6g}", len(top)])
