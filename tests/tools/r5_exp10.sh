#!/bin/bash
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(pwd)
out=$root/gpurun_out/r5j; mkdir -p $out
cd $root
for v in merge chained merge chained; do
  [ $v = chained ] && export PSS_BIG_MERGE=0
  timeout 600 python tests/tools/real_text.py 29 3 nocheck > $out/real_$v.txt 2>&1; grep "build" $out/real_$v.txt | tail -1 | cut -c1-60
  for c in source mixed dup_blocks; do
    timeout 600 python tests/tools/sa_perf.py $c 29 3 > $out/${c}_$v.txt 2>&1; tail -1 $out/${c}_$v.txt | cut -c1-70
  done
  unset PSS_BIG_MERGE
done
timeout 600 python tests/tools/real_text.py 29 2 > $out/real_check.txt 2>&1; tail -1 $out/real_check.txt
timeout 900 python -m pytest tests/test_sa_gpu.py -q -x -k "large_groups" > $out/pytest_sa.log 2>&1
tail -3 $out/pytest_sa.log
cd /tmp && export TMPDIR=/tmp; cd $root
