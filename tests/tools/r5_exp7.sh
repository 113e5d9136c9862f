#!/bin/bash
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(pwd)
out=$root/gpurun_out/r5g; mkdir -p $out
cd $root
timeout 1700 python -m pytest tests -q -m gpu --durations=25 > $out/pytest_gpu.log 2>&1
tail -45 $out/pytest_gpu.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err ) 2>&1 | tail -3
python - <<P
import json
d=json.loads(open('$out/bench_default.json').read().strip().splitlines()[-1])
print(json.dumps(d['summary'])[:1500])
print(json.dumps(d.get('real_files'))[:900])
print(json.dumps([a for a in d.get('adversarial') or [] if a['corpus']=='source'])[:900])
print(json.dumps(d.get('e2e',{}).get('striped_format_2'))[:600])
print(json.dumps(d.get('cpu_baseline'))[:600])
print(json.dumps(d.get('corpus15',{}).get('cpu_baseline'))[:900])
P
tail -5 $out/bench_default.err
