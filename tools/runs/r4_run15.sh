set -x
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "layernorm_fold" 2>&1 | tail -5
timeout 900 python tools/bench_unet_ab.py --what ln --rounds 2 2>&1 | tail -6 | tee gpurun_out/r4_unet_ab_ln_fold.log
