"""Local-sort kernel time of libpss variants (timing experiments): python tests/tools/msd_local_ab.py lib1.so lib2.so ..."""
import ast, os, subprocess, sys
for l in sys.argv[1:]:
    env = dict(os.environ, PSS_LIBPSS=os.path.abspath(l), PSS_PROFILE_ALL='1')
    out = subprocess.run([sys.executable, 'tests/tools/sa_perf.py', 'lines', '29', '4'], env=env, capture_output=True, text=True).stdout
    for line in out.splitlines():
        if line.startswith('rep 3'):
            d = ast.literal_eval(line[line.index('{'):])
            print(f"{os.path.basename(l):30s} total {d['ms_total']:.2f} g1 {d['msd_ms_g1']:.2f} g2 {d['msd_ms_g2']:.2f} local {d['msd_ms_local']:.2f}")
