"""Summarise rocprofv3 --pmc CSVs: per kernel name, mean counter value per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ''
agg = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(f'{root}/g*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name']
        if filt and filt not in name:
            continue
        agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
for name, cs in agg.items():
    print(name[:90])
    for c, v in sorted(cs.items()):
        v2 = sorted(v)
        print(f'   {c:28s} n={len(v):4d} mean={sum(v)/len(v):16.1f} max={v2[-1]:16.1f}')
