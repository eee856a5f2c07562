#!/bin/bash
# round 5, GPU call 2: XLV2 precise, fold model-level test, IPC self-test, chain per stage, config-1 CPU un-extrapolated
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_precise_gpu.py tests/test_llm_plain16_gpu.py tests/test_batched_decode_gpu.py tests/test_ipc_comm_gpu.py tests/test_tensor_parallel_gpu.py tests/test_models_gpu.py -q -s --timeout 600 > $O/r5_run2_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|FAILED|ERROR|RMSNorm fold at|ResamplerXLV2 " $O/r5_run2_tests.log | tail -30
timeout 900 python -m pytest tests/test_fulldepth_gpu.py -q -s --timeout 800 > $O/r5_fulldepth_full.log 2>&1; echo "fulldepth rc=$?"; grep -E "full depth|passed|failed" $O/r5_fulldepth_full.log | tail -30
timeout 900 python bench.py --config 1 --cpu-baseline full --steps 5 --warmup 2 > $O/r5_bench_config1_cpu_full.json 2> $O/r5_bench_config1_cpu_full.err; echo "config1 rc=$?"; tail -c 1500 $O/r5_bench_config1_cpu_full.json
