"""Aggregate a rocprofv3 counter_collection CSV per kernel name: launches, mean counter value."""
import csv
import sys
from collections import defaultdict

path, ctr = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: [0, 0.0])
with open(path) as f:
    for row in csv.DictReader(f):
        if row.get("Counter_Name") != ctr:
            continue
        name = row["Kernel_Name"]
        if "efg::" not in name:
            continue
        base = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        a = acc[base]
        a[0] += 1
        a[1] += float(row["Counter_Value"])
print("kernel,launches,mean_%s,total_%s" % (ctr, ctr))
for k, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("%s,%d,%.3f,%.3f" % (k.replace(",", ";"), n, v / n, v))
