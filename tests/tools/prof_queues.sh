#!/bin/bash
# Kernel trace of the builds of one corpus, every launch >= <min_us> with its queue: tests/tools/prof_queues.sh <corpus> <builds> <tag> [min_us=300]
c=${1:-source}; k=${2:-3}; tag=${3:-probe}; mn=${4:-300}
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(pwd)
ev=$root/gpurun_out/ev; mkdir -p $ev
cd /tmp && export TMPDIR=/tmp; cd $root
tool="tests/tools/sa_perf.py $c 29 $k"
[ $c = real ] && tool="tests/tools/real_text.py 29 $k nocheck"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ev/prof_$tag -o t -- python3 $tool > $ev/${tag}.log 2>&1
kt=$(ls $ev/prof_$tag/t_kernel_trace.csv $ev/prof_$tag/*/t_kernel_trace.csv 2>/dev/null | head -1)
python tests/tools/timeline.py $kt $mn q > $ev/${tag}_timeline_q.txt 2>&1
rm -rf $ev/prof_$tag
tail -2 $ev/${tag}.log | cut -c1-300
