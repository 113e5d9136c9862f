#!/bin/bash
# round 5, first GPU call: new tests + baselines of the real-text regime + the first experiments
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(pwd)
out=$root/gpurun_out/r5a; mkdir -p $out
cd $root
timeout 900 python -m pytest tests/test_rccl_faults_gpu.py tests/test_parity_gpu.py -q -x -k "rccl or faults or result_order or format_2 or gather or injected or failing or recv or go_no_go or hold_up" > $out/pytest_new.log 2>&1
tail -5 $out/pytest_new.log
PSS_TIMING=1 timeout 600 python tests/tools/real_text.py 29 2 > $out/real_base.txt 2>&1
tail -4 $out/real_base.txt
PSS_TIMING=1 PSS_ANCHOR_MIN_OMEGA=9 PSS_PROBE_SKIP_PCT=10 PSS_ANCHOR_CAP_DIV=4 timeout 600 python tests/tools/real_text.py 29 2 > $out/real_omega9.txt 2>&1
tail -4 $out/real_omega9.txt
PSS_TIMING=1 timeout 600 python tests/tools/sa_perf.py source 29 2 > $out/source_base.txt 2>&1
tail -2 $out/source_base.txt | cut -c1-400
PSS_TIMING=1 PSS_ANCHOR_MIN_OMEGA=9 PSS_PROBE_SKIP_PCT=10 PSS_ANCHOR_CAP_DIV=4 timeout 600 python tests/tools/sa_perf.py source 29 2 > $out/source_omega9.txt 2>&1
tail -2 $out/source_omega9.txt | cut -c1-400
timeout 300 python tests/tools/sa_perf.py source 24 2 check > $out/source_24_check.txt 2>&1
tail -2 $out/source_24_check.txt | cut -c1-200
