#!/bin/bash
# The counter files and the default bench line again, on the final sources of a round (round_evidence.sh without its
# long legs):   PSS_TREE_COMMIT=$(git rev-parse --short HEAD) tests/tools/round_pmc_refresh.sh r06
tag=${1:-r06}
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(pwd)
ev=$root/gpurun_out/ev; mkdir -p $ev/json
cd /tmp && export TMPDIR=/tmp; cd $root
timeout 600 python tests/tools/real_text.py 29 3 > $ev/${tag}_real_files.txt 2>&1
real_bytes=$(grep -o "^[0-9]* bytes of real files" $ev/${tag}_real_files.txt | head -1 | cut -d' ' -f1)
for spec in lines:3 words:2 dup_blocks:2 mixed:2 source:2 real:2 runs:2; do
  c=${spec%%:*}; k=${spec#*:}
  timeout 900 tests/tools/pmc_traffic.sh $ev/pmc_$c $c $k > /dev/null 2>&1
  if [ $c = real ]; then PSS_PMC_BYTES=$real_bytes python tests/tools/pmc_traffic_json.py $ev/pmc_$c $k $ev/json $c > $ev/pmc_$c.ratios.txt 2>&1
  else python tests/tools/pmc_traffic_json.py $ev/pmc_$c $k $ev/json $c > $ev/pmc_$c.ratios.txt 2>&1; fi
done
cp $ev/json/*.json $root/profiles/ 2>/dev/null
timeout 600 tests/tools/pmc_requests.sh $ev/pmcreq_words words 1 > /dev/null 2>&1
python tests/tools/pmc_requests_json.py $ev/pmcreq_words $ev/json/pmc_requests_words.json 1 > $ev/pmc_requests_words.txt 2>&1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $ev/${tag}_bench_default.json 2> $ev/bench_default.err
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
cp $ev/json/*.json $root/profiles/ 2>/dev/null
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $ev/${tag}_bench_default.json 2> $ev/bench_default.err
rm -rf $ev/pmc_runs $ev/pmc_lines $ev/pmc_words $ev/pmc_dup_blocks $ev/pmc_mixed $ev/pmc_source $ev/pmc_real $ev/pmcreq_words $ev/pmc_search
python tests/tools/check_evidence.py 2>&1 | tail -12
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
