#!/bin/bash
# Kernel trace of the builds of one corpus: tests/tools/prof_one.sh <corpus> <builds> <tag>  -> gpurun_out/ev/<tag>_{kernel_stats.csv,timeline.txt}
c=${1:-lines}; k=${2:-12}; tag=${3:-probe}
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(pwd)
ev=$root/gpurun_out/ev; mkdir -p $ev
cd /tmp && export TMPDIR=/tmp; cd $root
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ev/prof_$tag -o t -- python3 tests/tools/sa_perf.py $c 29 $k > $ev/${tag}.log 2>&1
cp $ev/prof_$tag/t_kernel_stats.csv $ev/${tag}_kernel_stats.csv
python tests/tools/timeline.py $ev/prof_$tag/t_kernel_trace.csv 100 > $ev/${tag}_timeline.txt 2>&1
rm -rf $ev/prof_$tag
tail -3 $ev/${tag}.log | cut -c1-200
cat $ev/${tag}_timeline.txt
