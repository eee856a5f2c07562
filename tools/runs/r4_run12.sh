set -x
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_batched_decode_gpu.py tests/test_models_gpu.py tests/test_golden_gpu.py -q -m gpu 2>&1 | tail -4
for b in 32 16 32; do
timeout 900 python bench.py --config 2 --batch $b --steps 4 --warmup 2 --also-dtype none --no-cpu-baseline 2>/dev/null > /tmp/line.json
python - <<PY
import json
d = json.load(open("/tmp/line.json"))
gr = d["roofline_phases"]["decode"]["graph_replay"]
print("config 2 batch $b: %.3f gens/s, %.1f ms per step, decode graph replay %.3f ms per token step = %.0f GB/s (%.3f of HBM peak)" % (d["value"], d["ms_per_step"], gr["ms_per_token"], gr["achieved"], gr["frac"]))
PY
cp /tmp/line.json gpurun_out/r4_bench_config2_batch$b.json
done 2>&1 | tee gpurun_out/r4_config2_batch32.log
