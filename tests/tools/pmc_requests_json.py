"""rocprofv3 request counters of tests/tools/pmc_requests.sh -> profiles/pmc_requests_<corpus>.json: per kernel the read
requests that reached the fabric (TCC_EA0_RDREQ), how many of them were 32-byte ones, TCC_BUBBLE (128-byte requests, where
the counter is wired) and the share that went to DRAM -- the calibration FETCH_SIZE needs for kernels that gather
(FETCH_SIZE = (BUBBLE x 128 + (RDREQ - BUBBLE - RDREQ_32B) x 64 + RDREQ_32B x 32) / 1024 by rocprofv3's own definition).

    python tests/tools/pmc_requests_json.py <dir> <out json> <builds>
"""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import csv
import glob
import json
import sys
from collections import defaultdict

root, out, builds = sys.argv[1], sys.argv[2], int(sys.argv[3])
agg = defaultdict(lambda: defaultdict(float))
launches = defaultdict(int)
for f in glob.glob(f'{root}/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].replace('void ', '').replace('pss::', '').split('(')[0]
        agg[name][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'TCC_EA0_RDREQ_sum':
            launches[name] += 1
res = {}
for name, cs in sorted(agg.items(), key=lambda kv: -kv[1].get('TCC_EA0_RDREQ_sum', 0)):
    rd = cs.get('TCC_EA0_RDREQ_sum', 0.0)
    if rd < 1e6:
        continue
    res[name] = {'launches_per_build': launches[name] / builds, 'RDREQ_per_build': int(rd / builds),
                 'RDREQ_32B_share': round(cs.get('TCC_EA0_RDREQ_32B_sum', 0.0) / rd, 4),
                 'BUBBLE_per_RDREQ': round(cs.get('TCC_BUBBLE_sum', 0.0) / rd, 4),
                 'DRAM_share': round(cs.get('TCC_EA0_RDREQ_DRAM_sum', 0.0) / rd, 4),
                 'bytes_at_64B_per_request': int(rd / builds * 64)}
json.dump({**__import__('tree_hash').stamp(), 'source': 'tests/tools/pmc_requests.sh (rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum '
                     'TCC_EA0_RDREQ_DRAM_sum, one run)', 'kernels': res}, open(out, 'w'), indent=1)
for k, v in list(res.items())[:12]:
    print(k, v)
