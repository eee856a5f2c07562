set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "fused_groupnorm" > gpurun_out/r4_gn_fused_test.log 2>&1; echo "gn test rc=$?"; tail -15 gpurun_out/r4_gn_fused_test.log
timeout 1500 python -m pytest tests/test_fullsize_gpu.py tests/test_fullsize2_gpu.py tests/test_adapter_e2e_gpu.py tests/test_golden_gpu.py tests/test_tensor_parallel_gpu.py -q -m gpu -x -s > gpurun_out/r4_unet_after_gn.log 2>&1; echo "unet rc=$?"; grep -E "rel-L2|passed|failed|Error" gpurun_out/r4_unet_after_gn.log | tail -30
timeout 900 python bench.py --steps 3 --warmup 1 --also-dtype none --no-cpu-baseline > gpurun_out/r4_bench_gn.json 2> gpurun_out/r4_bench_gn.err; echo "bench rc=$?"; cut -c1-330 gpurun_out/r4_bench_gn.json
