#!/bin/bash
# headline step at 16 vs 32 lock-step generations, same box (no roofline / CPU baseline passes)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
for b in 16 32; do
  timeout 420 python bench.py --batch $b --steps 2 --warmup 1 --also-dtype none --no-cpu-baseline --no-roofline > /tmp/l.json 2> /tmp/l.err; echo "batch $b rc=$?"
  python - <<PY
import json
try:
    d = json.load(open("/tmp/l.json")); print("batch $b: %.4f gens/s, %.1f ms per step" % (d["value"], d["ms_per_step"]))
except Exception as e:
    print("no line", e); print(open("/tmp/l.err").read()[-1500:])
PY
done 2>&1 | tee $O/r4_headline_batch_16_32.log
