#!/bin/bash
# Kernel statistics and timelines of one build per corpus (the part of round_evidence.sh that follows the build kernels):
#   tests/tools/round_kernel_stats.sh r04   -> gpurun_out/ev/<tag>_sa_build_*_kernel_stats.csv, <tag>_timeline_*.txt
tag=${1:-r04}
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(pwd)
ev=$root/gpurun_out/ev; mkdir -p $ev
cd /tmp && export TMPDIR=/tmp; cd $root
for spec in lines:12 words:5 dup_blocks:3 mixed:3; do
  c=${spec%%:*}; k=${spec#*:}
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ev/prof_$c -o t -- python3 tests/tools/sa_perf.py $c 29 $k > $ev/prof_$c.log 2>&1
  cp $ev/prof_$c/t_kernel_stats.csv $ev/${tag}_sa_build_${c}_kernel_stats.csv
  python tests/tools/timeline.py $ev/prof_$c/t_kernel_trace.csv 400 > $ev/${tag}_timeline_$c.txt 2>&1
  rm -rf $ev/prof_$c
done
ls -la $ev | tail -12
