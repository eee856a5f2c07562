#!/bin/bash
# round 5, GPU call 3: the config-0 chain at the real 50 steps, attention PMC, headline with precise LLM / XLV2 (+ request pipelining A/B)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_fulldepth_gpu.py -q -s --timeout 800 > $O/r5_fulldepth_full.log 2>&1; echo "fulldepth rc=$?"; grep -E "config-0|passed|failed" $O/r5_fulldepth_full.log | tail -14
for ov in 0 1 0 1; do
  timeout 600 python bench.py --steps 3 --warmup 1 --overlap $ov --also-dtype none --no-cpu-baseline --no-roofline > $O/r5_bench_overlap$ov.json 2> $O/r5_bench_overlap$ov.err; echo "bench overlap=$ov rc=$?"; python -c "
import json,sys
r=json.loads([l for l in open('$O/r5_bench_overlap$ov.json') if l.startswith('{')][-1]); print('overlap $ov', r['value'], r['ms_per_step'])"
done
bash tools/lab/attn_pmc.sh r5_attn 0 > $O/r5_attn_pmc.out 2>&1; echo "attn pmc rc=$?"; tail -50 $O/r5_attn_pmc_table.txt
