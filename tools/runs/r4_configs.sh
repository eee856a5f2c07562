#!/bin/bash
# round 4: BASELINE configs 1-5 (fp16, the default dtype), the latency launcher's functional run with ranks sharing one GPU
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
: > $O/r4_bench_configs.jsonl
for c in 1 2 3 4 5; do
  extra="--no-cpu-baseline"
  [ $c = 1 ] && extra="--cpu-baseline full"
  python bench.py --config $c --steps 3 --warmup 1 --also-dtype none $extra >> $O/r4_bench_configs.jsonl 2>> $O/r4_bench_configs.err
done
python - <<PY
import json
for l in open("$O/r4_bench_configs.jsonl"):
    d = json.loads(l)
    gr = (d.get("roofline_phases") or {}).get("decode", {}).get("graph_replay") or {}
    print(d["config"].get("baseline_config"), round(d["value"], 4), d["unit"], round(d["ms_per_step"], 1), (d.get("cpu_baseline") or {}).get("value"), gr.get("ms_per_token"), gr.get("frac"))
PY
for n in 2 4; do
  timeout 900 python tools/bench_tp_latency.py --gpus $n --share-gpu --steps 1 --warmup 1 --unet-steps 4 --unet-comm ipc > $O/r4_tp_launcher_share_gpu_$n.json 2> $O/r4_tp_launcher_share_gpu_$n.err; echo "tp $n rc=$?"; tail -c 900 $O/r4_tp_launcher_share_gpu_$n.json; echo; tail -3 $O/r4_tp_launcher_share_gpu_$n.err
done
