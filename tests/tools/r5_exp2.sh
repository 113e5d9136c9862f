#!/bin/bash
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(pwd)
out=$root/gpurun_out/r5b; mkdir -p $out
cd $root
timeout 900 python -m pytest tests/test_rccl_faults_gpu.py -q -x > $out/pytest_rccl.log 2>&1
tail -5 $out/pytest_rccl.log
timeout 900 python -m pytest tests/test_parity_gpu.py -q -x -k "result_order" > $out/pytest_order.log 2>&1
tail -5 $out/pytest_order.log
PSS_TIMING=1 timeout 600 python tests/tools/real_text.py 29 3 > $out/real_side.txt 2>&1
grep -v "^\[pss\]" $out/real_side.txt | tail -5
PSS_ANCHOR_SIDE=0 timeout 600 python tests/tools/real_text.py 29 3 > $out/real_noside.txt 2>&1
grep -v "^\[pss\]" $out/real_noside.txt | tail -4
timeout 600 python tests/tools/sa_perf.py source 29 3 > $out/source_side.txt 2>&1
tail -2 $out/source_side.txt | cut -c1-150
timeout 600 python tests/tools/sa_perf.py mixed 29 3 > $out/mixed_side.txt 2>&1
tail -2 $out/mixed_side.txt | cut -c1-150
timeout 600 python tests/tools/sa_perf.py words 29 3 > $out/words.txt 2>&1
tail -1 $out/words.txt | cut -c1-150
timeout 300 python tests/tools/sa_perf.py source 24 2 check > $out/source_24_check.txt 2>&1
tail -2 $out/source_24_check.txt | cut -c1-200
PSS_ANCHOR_SIDE=1 timeout 300 python tests/tools/anchor_check.py 60 9001 > $out/anchor_check_side.txt 2>&1
tail -2 $out/anchor_check_side.txt
