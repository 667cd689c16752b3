"""Summarise rocprofv3 --pmc CSV output per kernel: mean counter value per dispatch."""
import csv, glob, sys, collections
d = sys.argv[1]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(rows.items()):
    if not any(t in k for t in ("dw2", "pw", "scale", "unpack", "quantact")):
        continue
    print(k, "dispatches", len(next(iter(cs.values()))))
    for c, v in sorted(cs.items()):
        print("   %-28s mean %.4g" % (c, sum(v) / len(v)))
