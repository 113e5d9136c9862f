#!/bin/bash
# HBM traffic of whole builds by PMC, per corpus (separate FETCH_SIZE / WRITE_SIZE passes) -> gpurun_out/ev/json/*.json
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(pwd)
ev=$root/gpurun_out/ev; mkdir -p $ev/json
cd $root
for spec in lines:3 words:2 dup_blocks:2 mixed:2; do
  c=${spec%%:*}; k=${spec#*:}
  timeout 900 tests/tools/pmc_traffic.sh $ev/pmc_$c $c $k > /dev/null 2>&1
  python tests/tools/pmc_traffic_json.py $ev/pmc_$c $k $ev/json $c > $ev/pmc_$c.ratios.txt 2>&1
done
rm -rf $ev/pmc_lines $ev/pmc_words $ev/pmc_dup_blocks $ev/pmc_mixed
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $ev/r04_bench_default.json 2> $ev/bench_default.err
ls -la $ev/json
