#!/bin/bash
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(pwd)
out=$root/gpurun_out/r5l; mkdir -p $out/json
cd $root
export PSS_TREE_COMMIT=96f3504
timeout 900 tests/tools/pmc_traffic.sh $out/pmc_runs runs 2 > /dev/null 2>&1
python tests/tools/pmc_traffic_json.py $out/pmc_runs 2 $out/json runs > $out/pmc_runs.ratios.txt 2>&1
rm -rf $out/pmc_runs
cat $out/json/pmc_build_traffic_runs.json | head -12
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/r05_bench_default.json 2> $out/bench.err
python - <<P
import json
d=json.loads(open('$out/r05_bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['roofline']['traffic_evidence'])
print(d['real_files'])
P
timeout 400 python tests/tools/fuzz.py 300 9301 > $out/fuzz.txt 2>&1; tail -1 $out/fuzz.txt
timeout 300 python tests/tools/fuzz_search.py 150 9302 > $out/fuzz_search.txt 2>&1; tail -1 $out/fuzz_search.txt
