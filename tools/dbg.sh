for dbg in 0 1 2; do echo "debug=$dbg"; PSS_RS_DEBUG=$dbg PSS_SPARSE=1 PSS_KEY_CHARS=10 PSS_LIBPSS=$PWD/build/variants/libpss_dbg.so PSS_PROFILE_ALL=1 python tools/sa_perf.py lines 29 3 2>&1 | grep "^rep" | python -c "
import sys,ast
for l in sys.stdin:
    d=ast.literal_eval(l[l.index('{'):]); print('  pairs/launch %.3f ms  text %.3f' % (d['ms_pairs']/max(1,d['pairs_launches']), d['ms_text']))"; done
