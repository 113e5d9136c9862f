#!/bin/bash
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(pwd)
out=$root/gpurun_out/r5f; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $root
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o t -- python3 tests/tools/real_text.py 29 3 nocheck > $out/prof.log 2>&1
python tests/tools/timeline.py $out/prof/*/t_kernel_trace.csv 400 > $out/timeline_real.txt 2>&1 || python tests/tools/timeline.py $out/prof/t_kernel_trace.csv 400 > $out/timeline_real.txt 2>&1
grep "build" $out/prof.log | tail -2
grep -A22 "per kernel" $out/timeline_real.txt
rm -rf $out/prof
PSS_ANCHOR_SIDE=0 timeout 600 python tests/tools/real_text.py 29 3 nocheck > $out/real_noside.txt 2>&1; grep build $out/real_noside.txt | tail -1 | cut -c1-60
for c in source mixed dup_blocks words; do
timeout 600 python tests/tools/sa_perf.py $c 29 3 > $out/$c.txt 2>&1
tail -1 $out/$c.txt | cut -c1-100
done
timeout 600 python tests/tools/real_text.py 29 2 > $out/real_check.txt 2>&1; tail -1 $out/real_check.txt
PSS_TIMING=1 timeout 600 python tests/tools/real_e2e.py 29 2000 > $out/real_e2e.txt 2>&1
tail -1 $out/real_e2e.txt | cut -c1-420
timeout 900 python -m pytest tests/test_rccl_faults_gpu.py -q -x > $out/pytest_rccl.log 2>&1
tail -3 $out/pytest_rccl.log
timeout 1500 python -m pytest tests/test_sa_gpu.py -q -x > $out/pytest_sa.log 2>&1
tail -2 $out/pytest_sa.log
timeout 1500 python -m pytest tests/test_fullsize_gpu.py -q -x -k "largest or format_2 or general_repeats" --durations=10 > $out/pytest_big.log 2>&1
tail -18 $out/pytest_big.log
