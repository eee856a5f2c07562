#!/bin/bash
# configs 1, 2, 5 at HEAD (3 and 4: tools/runs/r4_run16.sh)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
: > $O/r4_bench_configs_125_head.jsonl
for c in 1 2 5; do
  timeout 200 python bench.py --config $c --steps 3 --warmup 1 --also-dtype none --no-cpu-baseline >> $O/r4_bench_configs_125_head.jsonl 2>> $O/r4_bench_configs_125_head.err; echo "config $c rc=$?"
done
python - <<PY
import json
for l in open("$O/r4_bench_configs_125_head.jsonl"):
    d = json.loads(l)
    gr = (d.get("roofline_phases") or {}).get("decode", {}).get("graph_replay") or {}
    print(d["config"].get("baseline_config"), round(d["value"], 4), d["unit"], round(d["ms_per_step"], 1), gr.get("ms_per_token"), gr.get("frac"))
PY
