"""Per-dispatch counter table for one kernel-name substring from a rocprofv3 counter_collection.csv."""
import csv, sys
from collections import OrderedDict
path, needle = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(path)))
d = OrderedDict()
for r in rows:
    if needle in r["Kernel_Name"]:
        d.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
        d[r["Dispatch_Id"]]["_grid"] = r.get("Grid_Size", "")
names = sorted({k for v in d.values() for k in v if k != "_grid"})
print("dispatch,grid," + ",".join(names))
for k, v in d.items():
    print(k + "," + str(v["_grid"]) + "," + ",".join(f"{v.get(n, 0):.4g}" for n in names))
