#!/bin/bash
# Everything profiles/ holds for a round, in one call on the GPU box (about 25 minutes):
#   PSS_TREE_COMMIT=$(git rev-parse --short HEAD) tests/tools/round_evidence.sh r06   -> gpurun_out/ev/*  (summaries only; copy what
#   is to be kept into profiles/).  The counter files are stamped with a hash of the engine's sources (tree_hash.py) and with
#   PSS_TREE_COMMIT (the box has no .git); the script ends with check_evidence.py on what it wrote.
tag=${1:-r06}
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(pwd)
ev=$root/gpurun_out/ev; mkdir -p $ev
cd $root
# 1. the GPU suite
timeout 2400 python -m pytest tests -q -m gpu -W error --durations=20 > $ev/${tag}_pytest_gpu.log 2>&1
tail -3 $ev/${tag}_pytest_gpu.log
# 2. the driver's command: below, once the counter files of THIS tree exist (the line quotes them and says whether they are current)
# 3. kernel stats under rocprofv3
cd /tmp && export TMPDIR=/tmp; cd $root
PSS_BENCH_NO_SECONDARY=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ev/prof_bench -o t -- python3 bench.py --no-cpu-baseline --no-corpus15 --no-e2e > $ev/${tag}_bench_under_rocprof.json 2>/dev/null
cp $ev/prof_bench/t_kernel_stats.csv $ev/${tag}_bench_kernel_stats.csv
for spec in lines:12 words:5 dup_blocks:3 mixed:3 source:3 real:3; do
  c=${spec%%:*}; k=${spec#*:}
  tool="tests/tools/sa_perf.py $c 29 $k"
  [ $c = real ] && tool="tests/tools/real_text.py 29 $k nocheck"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ev/prof_$c -o t -- python3 $tool > $ev/prof_$c.log 2>&1
  ks=$(ls $ev/prof_$c/t_kernel_stats.csv $ev/prof_$c/*/t_kernel_stats.csv 2>/dev/null | head -1)
  kt=$(ls $ev/prof_$c/t_kernel_trace.csv $ev/prof_$c/*/t_kernel_trace.csv 2>/dev/null | head -1)
  cp $ks $ev/${tag}_sa_build_${c}_kernel_stats.csv
  python tests/tools/timeline.py $kt 400 > $ev/${tag}_timeline_$c.txt 2>&1
  grep "build\|rep " $ev/prof_$c.log | tail -2 | cut -c1-200 >> $ev/${tag}_timeline_$c.txt
done
timeout 600 python tests/tools/real_text.py 29 3 > $ev/${tag}_real_files.txt 2>&1
timeout 600 python tests/tools/real_e2e.py 29 2000 > $ev/real_e2e.log 2>&1; tail -1 $ev/real_e2e.log > $ev/${tag}_real_files_e2e.json
# 4. HBM traffic by PMC (separate FETCH_SIZE / WRITE_SIZE passes)
mkdir -p $ev/json
real_bytes=$(grep -o "^[0-9]* bytes of real files" $ev/${tag}_real_files.txt | head -1 | cut -d' ' -f1)
for spec in lines:3 words:2 dup_blocks:2 mixed:2 source:2 real:2 runs:2; do
  c=${spec%%:*}; k=${spec#*:}
  timeout 900 tests/tools/pmc_traffic.sh $ev/pmc_$c $c $k > /dev/null 2>&1
  if [ $c = real ]; then PSS_PMC_BYTES=$real_bytes python tests/tools/pmc_traffic_json.py $ev/pmc_$c $k $ev/json $c > $ev/pmc_$c.ratios.txt 2>&1
  else python tests/tools/pmc_traffic_json.py $ev/pmc_$c $k $ev/json $c > $ev/pmc_$c.ratios.txt 2>&1; fi
done
cp $ev/json/*.json $root/profiles/ 2>/dev/null      # (on the box: the bench line below reads them)
# 4b. configs[2] / [3] on natural-text-like chunks, and the look-back micro-benchmark behind DESIGN 4.2
timeout 1500 python bench.py --config corpus15 --corpus words --qmin 16 --steps 2 --warmup 1 > $ev/${tag}_bench_corpus15_words.json 2> $ev/bench_corpus15_words.err
[ -x tests/tools/bin/lookback_micro ] && timeout 240 tests/tools/bin/lookback_micro > $ev/${tag}_lookback_micro_final.txt 2>&1
mkdir -p $ev/pmc_sq; tests/tools/pmc.sh $ev/pmc_sq/p -- python3 tests/tools/sa_perf.py lines 29 3 > /dev/null 2>&1
python tests/tools/pmc_summary.py $ev/pmc_sq/p msd_ > $ev/${tag}_pmc_sq_lines_kernels.txt 2>&1; rm -rf $ev/pmc_sq
cp $ev/${tag}_bench_corpus15_words.json $root/profiles/ 2>/dev/null
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $ev/${tag}_bench_default.json 2> $ev/bench_default.err
timeout 600 tests/tools/pmc_requests.sh $ev/pmcreq_words words 1 > /dev/null 2>&1
python tests/tools/pmc_requests_json.py $ev/pmcreq_words $ev/json/pmc_requests_words.json 1 > $ev/pmc_requests_words.txt 2>&1
timeout 1500 tests/tools/pmc_search.sh $ev/pmc_search > /dev/null 2>&1
msd=$(python - <<P
import json
try:
    d = json.loads(open('$ev/${tag}_bench_default.json').read().strip().splitlines()[-1])
    print(d['corpus15']['ms_device'])
except Exception:
    print('')
P
)
python tests/tools/pmc_search_json.py $ev/pmc_search $ev/json/pmc_search_corpus15.json $msd > $ev/pmc_search.txt 2>&1
# 5. the file API
python tests/tools/e2e_file.py 29 4 multi dir=/tmp check > $ev/${tag}_e2e_file_4x512MiB.txt 2>&1
python tests/tools/e2e_file.py 29 4 dir=/dev/shm >> $ev/${tag}_e2e_file_4x512MiB.txt 2>&1
python tests/tools/e2e_file.py 29 15 dir=/tmp > $ev/${tag}_e2e_file_15x512MiB.txt 2>&1
# 6. single-query latency against the number of results
python tests/tools/latency_hits.py 29 1 > $ev/lat1.txt 2>&1; tail -1 $ev/lat1.txt > $ev/${tag}_latency_vs_results_1chunk.json
python tests/tools/latency_hits.py 29 1 resident > $ev/lat1r.txt 2>&1; tail -1 $ev/lat1r.txt > $ev/${tag}_latency_vs_results_1chunk_low_latency.json
python tests/tools/latency_hits.py 29 15 > $ev/lat15.txt 2>&1; tail -1 $ev/lat15.txt > $ev/${tag}_latency_vs_results_15chunks.json
python tests/tools/latency_hits.py 29 15 resident > $ev/lat15r.txt 2>&1; tail -1 $ev/lat15r.txt > $ev/${tag}_latency_vs_results_15chunks_low_latency.json
# 7. fuzzing of the final code
# (every leg prints its exit code: a campaign must also END -- round 6's first final campaign aborted inside glibc and
#  `| tail -1` showed only the line of `timeout`)
leg() { "$@" 2>&1 | grep -v amdgpu.ids | tail -2; echo "exit ${PIPESTATUS[0]}"; }
(echo "# tests/tools/fuzz.py 300; FUZZ_BIG=1 fuzz.py 300; fuzz_search.py 200; anchor_check.py 150"
 leg timeout 500 python tests/tools/fuzz.py 300 7001
 FUZZ_BIG=1 leg timeout 500 python tests/tools/fuzz.py 300 7002
 leg timeout 900 python tests/tools/fuzz_search.py 200 7003
 leg timeout 300 python tests/tools/anchor_check.py 150 7004) > $ev/${tag}_fuzz.txt 2>&1
# keep the summaries only
rm -rf $ev/prof_* $ev/pmc_runs $ev/pmc_lines $ev/pmc_words $ev/pmc_dup_blocks $ev/pmc_mixed $ev/pmc_source $ev/pmc_real $ev/pmcreq_words $ev/pmc_search
# the counter files describe these sources?
mkdir -p $ev/chk/profiles; cp $ev/json/*.json $ev/chk/profiles/ 2>/dev/null
python - <<P
import glob, json, sys
sys.path.insert(0, 'tests/tools')
import tree_hash
now = tree_hash.csrc_hash()
bad = [f for f in glob.glob('$ev/json/pmc_*.json') if json.load(open(f)).get('csrc_sha256_16') != now]
print('evidence stamped with', now, '- stale:', bad)
P
ls -la $ev $ev/json
