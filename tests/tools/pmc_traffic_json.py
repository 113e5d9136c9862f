"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE CSVs (separate runs, tests/tools/pmc_traffic.sh) -> profiles/pmc_traffic.json and
profiles/pmc_build_traffic.json: HBM bytes per launch of every kernel of ONE default build of the lines 2^29 chunk.
bytes = FETCH_SIZE_KB * 1024 * 2 (gfx950 correction, /opt/skills/guides/MI355X_MICROARCH.md "HBM") + WRITE_SIZE_KB * 1024.

    python tests/tools/pmc_traffic_json.py <pmc dir with g*/pmc_counter_collection.csv> <builds in the run> <out dir> [corpus]

With a corpus other than lines only <out dir>/pmc_build_traffic_<corpus>.json is written.
"""
import csv, glob, json, os, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import tree_hash
root, builds, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
corpus = sys.argv[4] if len(sys.argv) > 4 else 'lines'
agg = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(f'{root}/g*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] in ('FETCH_SIZE', 'WRITE_SIZE'):
            agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
ALGO = {   # algorithmic bytes per element of the kernels of the initial sort
    'msd_scatter_kernel<true>': 9, 'msd_scatter_kernel<false>': 16, 'msd_local_fast_kernel': 12,
    'msd_scatter2_kernel<true, 1024>': 9, 'msd_scatter2_kernel<false, 1024>': 16,
    'msd_scatter2_kernel<true, 1024, true>': 9, 'msd_scatter_lb_kernel': 16,      # round 6: LSD order, second pass by look-back
    'msd_hist_raw_kernel': 2,
    'msd_hist_kernel<true>': 1, 'msd_hist_kernel<false>': 8,
    'fs_scatter_kernel<0, 4>': 9, 'fs_scatter_kernel<4, 4>': 16, 'fs_scatter_kernel<4, 0>': 12,
    # sample sort over 16-byte elements (ss_sort_impl.h)
    'ss_digits1_kernel': 3, 'ss_scatter_kernel<true>': 19, 'ss_digits2_kernel': 18, 'ss_scatter_kernel<false>': 34,
    'ss_local_kernel': 20, 'ss_local_seg_kernel': 20,
}
# Kernels whose reads are GATHERS (one 64-byte request per element: a random 4-byte rank, 32 bytes of text at a random
# offset): for them FETCH_SIZE is what it says.  Calibrated in round 4 with the request counters themselves
# (profiles/pmc_requests_words.json: text_keys makes 0.99 read requests per element, all of them counted at 64 B -- at
# 128 B each the kernel would have moved 6.7 TB/s, above what the chip sustains for streaming; ss_local, which streams
# 16 B per lane, makes one request per 128 B).  Every other kernel reads wide and coalesced: FETCH x 2 (guide, "HBM").
GATHER = ('text_keys_kernel', 'rank_keys_kernel', 'subset_keys_kernel', 'gather_names_kernel', 'build_keys_kernel',
          'probe_repeats_kernel', 'sample_keys_kernel', 'ss_sample_kernel')
n = int(os.environ.get('PSS_PMC_BYTES', 1 << 29))     # bytes of text per build (real files: what real_text.py collected)
kernels, total = {}, 0.0
for name, cs in agg.items():
    short = name.replace('void ', '').replace('pss::', '').split('(')[0]
    fk, wk = cs.get('FETCH_SIZE', []), cs.get('WRITE_SIZE', [])
    launches = max(len(fk), len(wk)) / builds
    f_b = sum(fk) / builds * 1024 * (1 if short.split('<')[0] in GATHER else 2)
    w_b = sum(wk) / builds * 1024
    total += f_b + w_b
    per = (f_b + w_b) / max(launches, 1)
    e = {'launches_per_build': launches, 'FETCH_SIZE_KB_per_launch': round(sum(fk) / max(len(fk), 1), 1),
         'WRITE_SIZE_KB_per_launch': round(sum(wk) / max(len(wk), 1), 1), 'bytes_per_launch': int(per)}
    if short in ALGO:
        e['algorithmic_bytes_per_launch'] = ALGO[short] * n
        e['ratio'] = round(per / (ALGO[short] * n), 3)
    kernels[short] = e
src = ('tests/tools/pmc_traffic.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate runs) around default-config '
       f'pss_sa_build_device calls on the {corpus} corpus, {n} bytes')
corr = ('bytes = FETCH_SIZE_KB * 1024 * 2 (gfx950 counts 64 B per 128-B request of a wide coalesced read) + WRITE_SIZE_KB * 1024; '
        'x 1 instead of x 2 for the gather kernels (' + ', '.join(GATHER) + '): their requests are 64 bytes, see profiles/pmc_requests_words.json')
big = {k: v for k, v in kernels.items() if v['bytes_per_launch'] * v['launches_per_build'] > 50e6}
if corpus == 'lines':
    json.dump(dict(tree_hash.stamp(), **{'source': src, 'correction': corr, 'kernels': big}), open(f'{out}/pmc_traffic.json', 'w'), indent=1)
json.dump(dict(tree_hash.stamp(), **{'source': src, 'correction': corr, 'text_bytes': n, 'total_bytes': int(total), 'bytes_per_suffix': round(total / n, 1),
           'by_kernel': {k: int(v['bytes_per_launch'] * v['launches_per_build']) for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]['bytes_per_launch'] * kv[1]['launches_per_build'])[:16]}}),
          open(f'{out}/pmc_build_traffic.json' if corpus == 'lines' else f'{out}/pmc_build_traffic_{corpus}.json', 'w'), indent=1)
print(json.dumps({k: v.get('ratio') for k, v in big.items()}))
