# LayerNorm fold (sx_gemm_ln): kernel test, complete-UNet test at 16 samples, same-process A/B of the 50-step loop
set -x
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "layernorm_fold or fused_groupnorm" 2>&1 | tail -15
timeout 1200 python -m pytest tests/test_fullsize2_gpu.py -q -m gpu -s -k "layernorm_fold" 2>&1 | tail -15
timeout 900 python tools/bench_unet_ab.py --what ln --rounds 2 2>&1 | tail -6 | tee gpurun_out/r4_unet_ab_ln_fold.log
