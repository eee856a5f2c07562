set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "fused_groupnorm" > gpurun_out/r4_gn_fused_test.log 2>&1; echo "gn test rc=$?"; tail -5 gpurun_out/r4_gn_fused_test.log
timeout 900 python tools/bench_unet_ab.py --rounds 3 > gpurun_out/r4_unet_ab_gn_fuse.log 2>&1; echo "ab rc=$?"; tail -4 gpurun_out/r4_unet_ab_gn_fuse.log
