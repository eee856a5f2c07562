set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fulldepth_gpu.py -x -q -m gpu -s > gpurun_out/r4_fulldepth.log 2>&1; echo "fulldepth rc=$?" 
timeout 600 python -m pytest tests/test_golden_gpu.py tests/test_ipc_comm_gpu.py tests/test_models_gpu.py tests/test_tensor_parallel_gpu.py -x -q -m gpu -s > gpurun_out/r4_golden_ipc.log 2>&1; echo "golden rc=$?"
tail -5 gpurun_out/r4_fulldepth.log; tail -5 gpurun_out/r4_golden_ipc.log
timeout 900 python bench.py --steps 2 --warmup 1 > gpurun_out/r4_bench_smoke.json 2> gpurun_out/r4_bench_smoke.err; echo "bench rc=$?"
cut -c1-600 gpurun_out/r4_bench_smoke.json
