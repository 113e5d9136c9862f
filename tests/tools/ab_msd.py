"""Interleaved A/B of libpss variants on the lines build: total ms and the MSD local sort's ms (profile mode).
usage: python tests/tools/ab_msd.py <rounds> lib1.so lib2.so ..."""
import ast
import os
import statistics
import subprocess
import sys

rounds = int(sys.argv[1])
libs = sys.argv[2:]
res = {l: [] for l in libs}
for _ in range(rounds):
    for l in libs:
        env = dict(os.environ, PSS_LIBPSS=os.path.abspath(l), PSS_PROFILE_ALL='1')
        out = subprocess.run([sys.executable, 'tests/tools/sa_perf.py', 'lines', '29', '5'], env=env, capture_output=True, text=True).stdout
        for line in out.splitlines():
            if line.startswith('rep') and not line.startswith('rep 0'):
                d = ast.literal_eval(line[line.index('{'):])
                res[l].append((d['ms_total'], d['msd_ms_local'], d['msd_ms_g1'], d['msd_ms_g2']))
for l in libs:
    if not res[l]:
        print(os.path.basename(l), 'no data')
        continue
    t = [x[0] for x in res[l]]
    lo = [x[1] for x in res[l]]
    print(f'{os.path.basename(l):24s} total min {min(t):6.3f} med {statistics.median(t):6.3f} | local min {min(lo):6.3f} med {statistics.median(lo):6.3f} '
          f'| g1 {min(x[2] for x in res[l]):5.3f} g2 {min(x[3] for x in res[l]):5.3f} (n={len(t)})')
