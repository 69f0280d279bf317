#!/usr/bin/env python
"""Per-kernel averages (over launches) of every counter in a rocprofv3 counter_collection.csv."""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in agg.values() for c in k})
print("kernel".ljust(40), "n".rjust(4), " ".join(n[-24:].rjust(24) for n in names))
rows = []
for k, c in agg.items():
    if k.startswith("at::") or k.startswith("__amd"):
        continue
    n = len(next(iter(c.values())))
    rows.append((k, n, [sum(c[x]) / len(c[x]) if x in c else 0.0 for x in names]))
rows.sort(key=lambda r: -max(r[2]) if r[2] else 0)
for k, n, v in rows[:24]:
    print(k[:40].ljust(40), str(n).rjust(4), " ".join(f"{x:24.4g}" for x in v))
