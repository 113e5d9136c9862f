#!/bin/bash
# Memory-side request counters per kernel (how many read requests of which size reach the fabric), one build:
#   tests/tools/pmc_requests.sh <out dir> [corpus=words] [builds=1]
out=$1; corpus=${2:-words}; builds=${3:-1}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum TCC_EA0_RDREQ_DRAM_sum --kernel-trace --output-format csv -d $out/r1 -o pmc -- python3 tests/tools/sa_perf.py $corpus 29 $builds > $out.r1.log 2>&1
ls $out/r1
