"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE CSVs of tests/tools/pmc_search.sh -> profiles/pmc_search_corpus15.json: bytes per
launch of the kernels of the batched search (the launches with the largest grid of every kernel: the 100 000-query batch
on 15 resident chunks), FETCH x 2 for the streaming kernels only (hit_lines / emit gather 64-byte sectors: raw).

    python tests/tools/pmc_search_json.py <dir with FETCH_SIZE/ and WRITE_SIZE/> <out json> [ms_device of the batch]
"""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import csv
import glob
import json
import sys
from collections import defaultdict

root, out = sys.argv[1], sys.argv[2]
ms = float(sys.argv[3]) if len(sys.argv) > 3 else None
NAMES = ('search_interval_lane_kernel', 'search_interval_group_kernel', 'search_interval_kernel', 'hit_lines_kernel', 'emit_kernel',
         'query_counts_kernel', 'scan_reduce_kernel', 'scan_apply_kernel')
vals = defaultdict(lambda: defaultdict(list))      # kernel -> counter -> [(grid, value)]
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    for f in glob.glob(f'{root}/{c}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != c:
                continue
            name = r['Kernel_Name'].replace('void ', '').replace('pss::', '').split('(')[0].split('<')[0]
            if name in NAMES:
                vals[name][c].append((int(r['Grid_Size']), float(r['Counter_Value'])))
kernels, total_raw, total_x2 = {}, 0.0, 0.0
for name, cs in vals.items():
    e = {}
    for c, rows in cs.items():
        gmax = max(g for g, _ in rows)
        big = [v for g, v in rows if g == gmax]
        e[c + '_KB_per_launch'] = round(sum(big) / len(big), 1)
        e['launches_averaged'] = len(big)
    f, w = e.get('FETCH_SIZE_KB_per_launch', 0.0) * 1024, e.get('WRITE_SIZE_KB_per_launch', 0.0) * 1024
    e['bytes_fetch_raw_plus_write'] = int(f + w)
    e['bytes_fetch_x2_plus_write'] = int(2 * f + w)
    kernels[name] = e
    total_raw += f + w
    total_x2 += 2 * f + w
res = {'source': 'tests/tools/pmc_search.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate runs) around '
                 'bench.py --config corpus15 --steps 1 (15 x 512 MiB resident, one batch of 100 000 queries of 4..32 bytes); per '
                 'kernel the launches with the largest grid',
       'note': 'FETCH_SIZE counts 64 B per read request; 128-byte requests of wide coalesced loads are tallied at half their size '
               '(guide: x 2 for streaming reads), sector gathers are not -- both totals are given; the truth lies between them',
       'kernels': kernels, 'total_bytes_raw': int(total_raw), 'total_bytes_fetch_x2': int(total_x2)}
if ms:
    res['ms_device_of_the_batch'] = ms
    res['achieved_gbs_raw'] = round(total_raw / ms / 1e6, 1)
    res['achieved_gbs_fetch_x2'] = round(total_x2 / ms / 1e6, 1)
res.update(__import__('tree_hash').stamp())
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps({k: v['bytes_fetch_raw_plus_write'] for k, v in kernels.items()}), res.get('achieved_gbs_raw'))
