#!/bin/bash
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(pwd)
out=$root/gpurun_out/r5i; mkdir -p $out
cd $root
for v in merge chained; do
  [ $v = chained ] && export PSS_NO_BIG_MERGE=1
  timeout 600 python tests/tools/real_text.py 29 3 > $out/real_$v.txt 2>&1; grep "build\|equal" $out/real_$v.txt | tail -3 | cut -c1-150
  for c in source mixed dup_blocks words; do
    timeout 600 python tests/tools/sa_perf.py $c 29 3 > $out/${c}_$v.txt 2>&1; tail -1 $out/${c}_$v.txt | cut -c1-70
  done
  unset PSS_NO_BIG_MERGE
done
timeout 1500 python -m pytest tests/test_sa_gpu.py -q -x > $out/pytest_sa.log 2>&1
tail -3 $out/pytest_sa.log
timeout 300 python tests/tools/anchor_check.py 120 9201 > $out/anchor_check.txt 2>&1
tail -1 $out/anchor_check.txt
timeout 400 python tests/tools/fuzz.py 200 9202 > $out/fuzz.txt 2>&1
tail -1 $out/fuzz.txt
FUZZ_BIG=1 timeout 400 python tests/tools/fuzz.py 200 9203 > $out/fuzz_big.txt 2>&1
tail -1 $out/fuzz_big.txt
timeout 1500 python -m pytest tests/test_fullsize_gpu.py -q -x -k "general_repeats or full_chunk or largest" --durations=8 > $out/pytest_big.log 2>&1
tail -14 $out/pytest_big.log
