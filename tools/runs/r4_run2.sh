set -x
mkdir -p gpurun_out
timeout 600 tools/lab/attn_lab 5 0 17 18 19 20 24 26 32 33 34 35 > gpurun_out/r4_attn_lab_opt_variants.log 2>&1; echo "attn_lab rc=$?"
grep -E "^==|median|BAD|LAB" gpurun_out/r4_attn_lab_opt_variants.log | head -120
timeout 900 python -m pytest tests/test_fulldepth_gpu.py -q -m gpu -s > gpurun_out/r4_fulldepth.log 2>&1; echo "fulldepth rc=$?"
grep -E "full depth|step |passed|failed|Error" gpurun_out/r4_fulldepth.log | head -40
timeout 900 python -m pytest tests/test_ipc_comm_gpu.py -q -m gpu -s > gpurun_out/r4_ipc_248.log 2>&1; echo "ipc rc=$?"
grep -E "RANK|passed|failed|Error|assert" gpurun_out/r4_ipc_248.log | head -40
