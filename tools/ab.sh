#!/bin/bash
# A/B of libpss variants: tools/ab.sh <corpus> <logn> lib1.so lib2.so ...
corpus=$1; logn=$2; shift 2
for rep in 1 2; do
for lib in "$@"; do
  echo -n "$(basename $lib): "
  PSS_LIBPSS=$PWD/$lib python tools/sa_perf.py $corpus $logn 3 2>&1 | tail -1 | python -c "
import sys,re,ast
l=sys.stdin.read()
d=ast.literal_eval(l[l.index('{'):])
print('total %.2f ms  pairs %.3f ms/launch (%d)  text %.3f ms  passes %d' % (d['ms_total'], d['ms_pairs']/max(1,d['pairs_launches']), d['pairs_launches'], d['ms_text'], d['initial_passes']))"
done; done
