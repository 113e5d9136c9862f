#!/bin/bash
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(pwd)
out=$root/gpurun_out/r5e; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $root
for v in merge nomerge; do
  [ $v = nomerge ] && export PSS_NO_MID_MERGE=1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$v -o t -- python3 tests/tools/real_text.py 29 3 nocheck > $out/prof_$v.log 2>&1
  python tests/tools/timeline.py $out/prof_$v/*/t_kernel_trace.csv 400 > $out/timeline_real_$v.txt 2>&1 || python tests/tools/timeline.py $out/prof_$v/t_kernel_trace.csv 400 > $out/timeline_real_$v.txt 2>&1
  grep "build" $out/prof_$v.log | tail -2
  tail -45 $out/timeline_real_$v.txt
  unset PSS_NO_MID_MERGE
done
rm -rf $out/prof_merge $out/prof_nomerge
PSS_TIMING=1 timeout 600 python tests/tools/real_e2e.py 29 2000 > $out/real_e2e.txt 2>&1
grep "\[pss\] \(writer\|add_file\|build\|record\)" $out/real_e2e.txt | tail -24; tail -1 $out/real_e2e.txt | cut -c1-300
timeout 900 python -m pytest tests/test_parity_gpu.py -q -x -k "striped or mapping or ingest or format_2" > $out/pytest_new.log 2>&1
tail -5 $out/pytest_new.log
timeout 900 python -m pytest tests/test_rccl_faults_gpu.py -q -x > $out/pytest_rccl.log 2>&1
tail -3 $out/pytest_rccl.log
