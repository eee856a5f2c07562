#!/bin/bash
# configs 3 and 4 (text -> image, edit with 3-way guidance) at HEAD: the LayerNorm fold on CFG batches 32 and 48
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
: > $O/r4_bench_configs_34_head.jsonl
for c in 3 4; do
  timeout 400 python bench.py --config $c --steps 2 --warmup 1 --also-dtype none --no-cpu-baseline --no-roofline >> $O/r4_bench_configs_34_head.jsonl 2>> $O/r4_bench_configs_34_head.err; echo "config $c rc=$?"
done
python - <<PY
import json
for l in open("$O/r4_bench_configs_34_head.jsonl"):
    d = json.loads(l)
    print(d["config"].get("baseline_config"), round(d["value"], 4), d["unit"], round(d["ms_per_step"], 1))
PY
tail -3 $O/r4_bench_configs_34_head.err
