"""Interleaved A/B of libpss variants (one process per (lib, round); min / median over rounds).
usage: python tests/tools/ab.py <corpus> <logn> <rounds> lib1.so lib2.so ..."""
import ast
import os
import statistics
import subprocess
import sys

corpus, logn, rounds = sys.argv[1], sys.argv[2], int(sys.argv[3])
libs = sys.argv[4:]
res = {l: [] for l in libs}
for _ in range(rounds):
    for l in libs:
        env = dict(os.environ, PSS_LIBPSS=os.path.abspath(l), PSS_PROFILE_ALL='1')
        out = subprocess.run([sys.executable, 'tests/tools/sa_perf.py', corpus, logn, '4'], env=env, capture_output=True, text=True).stdout
        for line in out.splitlines():
            if line.startswith('rep') and not line.startswith('rep 0'):
                d = ast.literal_eval(line[line.index('{'):])
                res[l].append((d['ms_total'], d['ms_pairs'] / max(1, d['pairs_launches']), d['ms_text']))
for l in libs:
    t = [x[0] for x in res[l]]
    p = [x[1] for x in res[l]]
    x = [x[2] for x in res[l]]
    print(f'{os.path.basename(l):28s} total min {min(t):7.2f} med {statistics.median(t):7.2f} | pairs/launch min {min(p):6.3f} med {statistics.median(p):6.3f} | text min {min(x):6.3f}  (n={len(t)})')
