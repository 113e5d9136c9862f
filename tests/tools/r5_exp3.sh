#!/bin/bash
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(pwd)
out=$root/gpurun_out/r5c; mkdir -p $out
cd $root
PSS_TIMING=1 timeout 600 python tests/tools/real_text.py 29 4 > $out/real_side2.txt 2>&1
grep -v "^\[pss\]" $out/real_side2.txt | tail -6
timeout 600 python tests/tools/sa_perf.py source 29 4 > $out/source_side2.txt 2>&1
tail -3 $out/source_side2.txt | cut -c1-150
timeout 600 python tests/tools/sa_perf.py dup_blocks 29 3 > $out/dup.txt 2>&1
tail -1 $out/dup.txt | cut -c1-150
timeout 3000 python -m pytest tests -q -m gpu -x --durations=15 > $out/pytest_gpu.log 2>&1
tail -30 $out/pytest_gpu.log
