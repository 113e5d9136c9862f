#!/bin/bash
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(pwd)
out=$root/gpurun_out/r5d; mkdir -p $out
cd $root
PSS_TIMING=1 timeout 600 python tests/tools/real_text.py 29 3 > $out/real.txt 2>&1
grep -v "^\[pss\]" $out/real.txt | tail -4
grep "text round\|rank round: h=20" $out/real.txt | tail -7
for c in source mixed dup_blocks words; do
timeout 600 python tests/tools/sa_perf.py $c 29 3 > $out/$c.txt 2>&1
tail -1 $out/$c.txt | cut -c1-120
done
timeout 1500 python -m pytest tests/test_sa_gpu.py -q -x > $out/pytest_sa.log 2>&1
tail -3 $out/pytest_sa.log
timeout 300 python tests/tools/anchor_check.py 100 9101 > $out/anchor_check.txt 2>&1
tail -1 $out/anchor_check.txt
timeout 400 python tests/tools/fuzz.py 200 9102 > $out/fuzz.txt 2>&1
tail -1 $out/fuzz.txt
timeout 1500 python -m pytest tests/test_fullsize_gpu.py -q -x -k "largest or format_2 or general_repeats" --durations=10 > $out/pytest_big.log 2>&1
tail -18 $out/pytest_big.log
