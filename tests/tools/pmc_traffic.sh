#!/bin/bash
# FETCH_SIZE and WRITE_SIZE (separate runs) for one default-config build of the bench chunk.
out=$1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/g1 -o pmc -- python3 tests/tools/sa_perf.py lines 29 1 > $out.g1.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/g2 -o pmc -- python3 tests/tools/sa_perf.py lines 29 1 > $out.g2.log 2>&1
python tests/tools/pmc_summary.py $out scatter
