#!/bin/bash
# round 3 final pass A: GPU test suite, headline line (bf16 + fp16), the other BASELINE configs
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > $O/r3_pytest_gpu.log; tail -3 $O/r3_pytest_gpu.log
python bench.py --steps 2 --warmup 1 > $O/r3_bench_line.json 2> $O/r3_bench_line.err; head -c 300 $O/r3_bench_line.json; echo
python bench.py --steps 2 --warmup 1 --dtype fp16 --no-cpu-baseline > $O/r3_bench_line_fp16.json 2>> $O/r3_bench_line.err; head -c 300 $O/r3_bench_line_fp16.json; echo
: > $O/r3_bench_configs.jsonl
for c in 1 2 3 4 5; do
  extra="--no-cpu-baseline"
  [ $c = 1 ] && extra="--cpu-baseline full"
  python bench.py --config $c --steps 2 --warmup 1 $extra >> $O/r3_bench_configs.jsonl 2>> $O/r3_bench_line.err
done
python - <<PY
import json
for l in open("$O/r3_bench_configs.jsonl"):
    d = json.loads(l)
    print(d["config"].get("baseline_config"), d["value"], d["unit"], d["ms_per_step"], (d.get("cpu_baseline") or {}).get("value"))
PY
