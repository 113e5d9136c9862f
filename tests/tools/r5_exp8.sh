#!/bin/bash
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(pwd)
out=$root/gpurun_out/r5h; mkdir -p $out
cd $root
for v in prio noprio prio noprio; do
  [ $v = noprio ] && export PSS_HELPER_PRIORITY=0
  timeout 600 python tests/tools/real_text.py 29 4 nocheck > $out/real_$v.txt 2>&1; grep build $out/real_$v.txt | tail -2 | cut -c1-150
  timeout 600 python tests/tools/sa_perf.py source 29 4 > $out/source_$v.txt 2>&1; tail -1 $out/source_$v.txt | cut -c1-70
  unset PSS_HELPER_PRIORITY
done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
