#!/bin/bash
# round 6: BASELINE configs 1-5 (fp16, the default dtype; config 2 also in bf16 = BASELINE's wording; config 2's line carries the
# 16-sequence precise run as `value` and the 32-sequence plain run as `value_plain16_batch32`), the latency launcher's functional run
# with ranks sharing one GPU (IPC collectives incl. their start-up self-test)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
: > $O/r6_bench_configs.jsonl
for c in 1 2 2b 3 4 5; do
  extra="--no-cpu-baseline"
  cc=$c
  [ $c = 2b ] && { cc=2; extra="$extra --dtype bf16"; }
  timeout 900 python bench.py --config $cc --steps 3 --warmup 1 --also-dtype none $extra >> $O/r6_bench_configs.jsonl 2>> $O/r6_bench_configs.err
done
python - <<PY
import json
for l in open("$O/r6_bench_configs.jsonl"):
    d = json.loads(l)
    gr = (d.get("roofline_phases") or {}).get("decode", {}).get("graph_replay") or {}
    print(d["config"].get("baseline_config"), d["dtype"], round(d["value"], 4), d["unit"], round(d["ms_per_step"], 1), gr.get("ms_per_token"), gr.get("frac"),
          "plain32:", d.get("value_plain16_batch32"), (d["config"].get("llm_mode") or "")[:24])
PY
for n in 2 4; do
  timeout 900 python tools/bench_tp_latency.py --gpus $n --share-gpu --steps 1 --warmup 1 --unet-steps 4 --unet-comm ipc > $O/r6_tp_launcher_share_gpu_$n.json 2> $O/r6_tp_launcher_share_gpu_$n.err; echo "tp $n rc=$?"; tail -c 600 $O/r6_tp_launcher_share_gpu_$n.json; echo; tail -3 $O/r6_tp_launcher_share_gpu_$n.err
done
