"""Tabulate a rocprofv3 --pmc counter_collection CSV: one row per dispatch (or per kernel name with
a substring filter: mean over dispatches), counters as columns."""
import collections, csv, glob, os, sys

d = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else None
paths = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
rows = []
for p in paths:
    rows += list(csv.DictReader(open(p)))
by = collections.OrderedDict()
for r in rows:
    if filt and filt not in r["Kernel_Name"]:
        continue
    key = (r["Dispatch_Id"], r["Kernel_Name"][:70], r.get("Grid_Size", "?"))
    by.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
names = sorted({c for v in by.values() for c in v})
print("dispatch kernel grid " + " ".join(names))
for (did, kn, g), v in sorted(by.items(), key=lambda kv: int(kv[0][0])):
    print(did, kn.replace(" ", "_"), g, " ".join("%.6g" % v.get(c, float("nan")) for c in names))
