set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_batched_decode_gpu.py tests/test_models_gpu.py tests/test_golden_gpu.py -q -m gpu -x > gpurun_out/r4_decode_fold_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r4_decode_fold_tests.log
for f in 1 0 1 0; do
  SX_RMS_FOLD=$f timeout 600 python bench.py --config 2 --steps 3 --warmup 1 --also-dtype none --no-cpu-baseline 2>/dev/null > /tmp/line.json
  python - <<PY
import json
d = json.load(open("/tmp/line.json"))
gr = d["roofline_phases"]["decode"]["graph_replay"]
print("SX_RMS_FOLD=$f: %.3f gens/s, %.1f ms per step, decode graph replay %.3f ms/token = %.0f GB/s (%.3f of HBM peak)" % (d["value"], d["ms_per_step"], gr["ms_per_token"], gr["achieved"], gr["frac"]))
PY
done 2>&1 | tee gpurun_out/r4_config2_rms_fold_ab.log
