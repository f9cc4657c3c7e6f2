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
scol = next((c for c in cols if "tart" in c and "imestamp" in c), None)  # dispatch duration UNDER the counter pass,
ecol = next((c for c in cols if "nd_" in c and "imestamp" in c), None)    # when rocprofv3 reports it
per = defaultdict(lambda: defaultdict(float))  # (kernel, counter) -> dispatch -> value (summed over XCDs / SEs)
dur = defaultdict(dict)                        # (kernel, counter) -> dispatch -> ns
for i, r in enumerate(rows):
    k = r[kcol].split("(")[0][:90]
    did = r[dcol] if dcol else i
    per[(k, r[ccol])][did] += float(r[vcol])
    if scol and ecol:
        try:
            dur[(k, r[ccol])][did] = float(r[ecol]) - float(r[scol])
        except ValueError:
            pass
w = csv.writer(sys.stdout)
w.writerow(["kernel", "counter", "sum", "launches", "per_launch", "top_mean", "top_n", "top_ms"])
for (k, c), d in sorted(per.items(), key=lambda kv: -sum(kv[1].values())):
    vals = list(d.values())
    v, n, mx = sum(vals), len(vals), max(vals)
    top_ids = [i for i, x in d.items() if x >= 0.5 * mx] if mx > 0 else list(d)
    top = [d[i] for i in top_ids]
    tms = [dur[(k, c)][i] for i in top_ids if i in dur[(k, c)]]
    w.writerow([k, c, f"{v:.6g}", n, f"{v / n:.6g}", f"{sum(top) / len(top):.6g}", len(top),
                f"{sum(tms) / len(tms) / 1e6:.6g}" if tms else ""])
