#!/bin/bash
# FETCH_SIZE and WRITE_SIZE (separate runs) for default-config builds of one 2^29 chunk:
#   tests/tools/pmc_traffic.sh <out dir> [corpus=lines] [builds=1]
out=$1; corpus=${2:-lines}; builds=${3:-1}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tool="tests/tools/sa_perf.py $corpus 29 $builds"
[ "$corpus" = real ] && tool="tests/tools/real_text.py 29 $builds nocheck"      # real files found on the machine (one chunk of them)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/g1 -o pmc -- python3 $tool > $out.g1.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/g2 -o pmc -- python3 $tool > $out.g2.log 2>&1
python tests/tools/pmc_summary.py $out scatter
