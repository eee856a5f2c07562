set -x
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "rmsnorm_fold" 2>&1 | tail -12
timeout 600 python -m pytest tests/test_batched_decode_gpu.py tests/test_models_gpu.py -q -m gpu -x 2>&1 | tail -3
for f in 1 0 1 0; do
  SX_GEMV_BAL20=$f timeout 600 python bench.py --config 2 --steps 3 --warmup 1 --also-dtype none --no-cpu-baseline 2>/dev/null > /tmp/line.json
  python - <<PY
import json
d = json.load(open("/tmp/line.json"))
gr = d["roofline_phases"]["decode"]["graph_replay"]
print("SX_GEMV_BAL20=$f: %.3f gens/s, %.1f ms per step, decode graph replay %.3f ms/token = %.0f GB/s (%.3f of HBM peak)" % (d["value"], d["ms_per_step"], gr["ms_per_token"], gr["achieved"], gr["frac"]))
PY
done 2>&1 | tee gpurun_out/r4_config2_bal20_ab.log
