set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fulldepth_gpu.py -q -m gpu -s -k config0 > gpurun_out/r4_fulldepth_chain.log 2>&1; echo "chain rc=$?"
grep -E "full depth|passed|failed|Error" gpurun_out/r4_fulldepth_chain.log | head -20
bash tools/runs/r4_pmc_unet.sh fp16
