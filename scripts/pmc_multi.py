"""Per-kernel means of several counters from a rocprofv3 counter_collection CSV."""
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for row in csv.DictReader(open(sys.argv[1])):
    n = row["Kernel_Name"]
    if "efg::" not in n: continue
    base = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    a = acc[base][row["Counter_Name"]]; a[0] += 1; a[1] += float(row["Counter_Value"])
ctrs = sorted({c for k in acc.values() for c in k})
print("kernel," + ",".join(ctrs))
for k, v in sorted(acc.items()):
    print(k.replace(",", ";") + "," + ",".join("%.4g" % (v[c][1] / max(v[c][0], 1)) for c in ctrs))
