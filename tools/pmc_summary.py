"""Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per kernel over the last N dispatches."""
import csv, sys, collections
import numpy as np
path, last = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 200
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(path)):
    agg[r['Kernel_Name'][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    if not k.startswith(("void k_", "k_")):
        continue
    print(k)
    for c, x in sorted(v.items()):
        print("   %-28s %14.1f  (n=%d)" % (c, np.mean(x[-last:]), len(x)))
