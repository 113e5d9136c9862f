#!/bin/bash
# FETCH_SIZE and WRITE_SIZE (separate runs) of the kernels of the 100 000-query batch on the 15-chunk corpus:
#   tests/tools/pmc_search.sh <out dir>
out=$1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$c -o pmc -- python3 bench.py --config corpus15 --steps 1 --warmup 0 --no-cpu-baseline > $out.$c.log 2>&1
done
ls $out
