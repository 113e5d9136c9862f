#!/bin/bash
# A/B of environment switches on one corpus: tests/tools/ab_env.sh <corpus> "<ENV=1 ...>" "<ENV=0 ...>" ...   (msd timings of the last of 4 builds)
c=$1; shift
for e in "$@"; do
  for rep in 1 2; do
    out=$(env $e PSS_PROFILE_ALL=1 python3 tests/tools/sa_perf.py $c 29 4 2>/dev/null | grep "^rep 3")
    python3 - "$e" "$out" <<'PY'
import ast, sys
e, line = sys.argv[1], sys.argv[2]
d = ast.literal_eval(line[line.index('{'):])
print(f"{e:40s} total {d['ms_total']:.2f} p1 {d['msd_ms_g1']:.2f} p2 {d['msd_ms_g2']:.2f} local {d['msd_ms_local']:.2f}")
PY
  done
done
