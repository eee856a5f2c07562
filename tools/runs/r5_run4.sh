#!/bin/bash
# round 5, GPU call: precise decode fold (gamma on the activation side), plane outputs from the GLU epilogue, decode tiles for every G,
# 8-row attention blocks for prefill — tests, same-box LLM phase A/B, config 5
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 1200 python -m pytest tests/test_precise_gpu.py tests/test_llm_plain16_gpu.py tests/test_batched_decode_gpu.py tests/test_models_gpu.py tests/test_golden_gpu.py tests/test_tensor_parallel_gpu.py -q -s --timeout 600 > $O/r5_run4_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|FAILED|ERROR|RMSNorm fold at|precise RMSNorm fold f" $O/r5_run4_tests.log | tail -30
timeout 600 python tools/bench_llm_precise_ab.py > $O/r5_llm_precise_ab2.log 2>&1; echo "ab rc=$?"; tail -5 $O/r5_llm_precise_ab2.log
timeout 900 python bench.py --config 5 --steps 3 --warmup 1 --also-dtype none --no-cpu-baseline > $O/r5_bench_config5.json 2> $O/r5_bench_config5.err; echo "config5 rc=$?"; python -c "
import json
r=json.loads([l for l in open('$O/r5_bench_config5.json') if l.startswith('{')][-1]); print('config 5', r['value'], r['ms_per_step'], (r.get('roofline_phases') or {}).get('decode',{}).get('graph_replay'))"
timeout 900 python -m pytest tests/test_fulldepth_gpu.py -q -s --timeout 800 -k llama > $O/r5_fulldepth_llm2.log 2>&1; echo "fulldepth llm rc=$?"; grep -E "full depth|passed|failed" $O/r5_fulldepth_llm2.log | tail -12
