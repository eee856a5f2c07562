#!/bin/bash
# round 5, GPU call 1: the precise LLM mode — kernel + model tests, full-depth parity, same-box cost A/B
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_precise_gpu.py -q -s --timeout 300 > $O/r5_precise_tests.log 2>&1; echo "precise tests rc=$?"; tail -25 $O/r5_precise_tests.log
timeout 900 python -m pytest tests/test_fulldepth_gpu.py -q -s --timeout 600 -k "llama" > $O/r5_fulldepth_llm.log 2>&1; echo "fulldepth llm rc=$?"; grep -v "^$" $O/r5_fulldepth_llm.log | tail -25
timeout 600 python tools/bench_llm_precise_ab.py > $O/r5_llm_precise_ab.log 2>&1; echo "ab rc=$?"; tail -6 $O/r5_llm_precise_ab.log
