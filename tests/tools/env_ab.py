"""Interleaved A/B of one environment switch: python tests/tools/env_ab.py VAR corpus logn rounds"""
import ast, os, statistics, subprocess, sys
var, corpus, logn, rounds = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
res = {'unset': [], 'set': []}
for _ in range(rounds):
    for mode in ('unset', 'set'):
        env = dict(os.environ)
        env.pop(var, None)
        if mode == 'set':
            env[var] = '1'
        out = subprocess.run([sys.executable, 'tests/tools/sa_perf.py', corpus, logn, '4'], env=env, capture_output=True, text=True).stdout
        for line in out.splitlines():
            if line.startswith('rep') and not line.startswith('rep 0'):
                res[mode].append(ast.literal_eval(line[line.index('{'):])['ms_total'])
for mode in res:
    t = res[mode]
    print(f'{var} {mode:6s}: min {min(t):.2f} med {statistics.median(t):.2f} ms (n={len(t)})')
