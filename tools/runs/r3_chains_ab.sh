#!/bin/bash
# same-box A/B: one vs two concurrent kernel chains in the denoise loop of the headline step
R=$GRAFT_REPO_ROOT
cd $R
for c in 2 1 2 1; do
  python bench.py --steps 2 --warmup 1 --chains $c --no-cpu-baseline --no-roofline 2>/dev/null > /tmp/line.json
  python - <<PY
import json
d = json.load(open("/tmp/line.json"))
print("chains $c:", round(d["value"], 4), "gens/s,", round(d["ms_per_step"], 1), "ms per step")
PY
done
