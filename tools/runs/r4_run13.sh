# decode attention with look-ahead K/V loads: tests, then same-box A/B against the previous library (config 2, batch 16 and 32)
set -x
L=seed-x_amd/lib
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_batched_decode_gpu.py tests/test_models_gpu.py tests/test_golden_gpu.py -q -m gpu 2>&1 | tail -4
cp $L/libseedx_hip.so /tmp/new.so
{
for v in prev new prev new; do
  if [ $v = prev ]; then cp $L/libseedx_hip_prev.so $L/libseedx_hip.so; else cp /tmp/new.so $L/libseedx_hip.so; fi
  for b in 16 32; do
    timeout 900 python bench.py --config 2 --batch $b --steps 4 --warmup 2 --also-dtype none --no-cpu-baseline 2>/dev/null > /tmp/line.json
    python - <<PY
import json
d = json.load(open("/tmp/line.json"))
gr = d["roofline_phases"]["decode"]["graph_replay"]
print("$v config 2 batch $b: %.3f gens/s, %.1f ms per step, decode graph replay %.3f ms per token step = %.0f GB/s (%.3f of HBM peak)" % (d["value"], d["ms_per_step"], gr["ms_per_token"], gr["achieved"], gr["frac"]))
PY
  done
done
} 2>&1 | grep -v "^+" | tee gpurun_out/r4_config2_attn_lookahead_ab.log
cp /tmp/new.so $L/libseedx_hip.so
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/ks -o run -- python $GRAFT_REPO_ROOT/bench.py --config 2 --batch 16 --steps 3 --warmup 1 --also-dtype none --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/kstats_top.py /tmp/ks 14 > gpurun_out/r4_cfg2_kstats_lookahead.txt 2>&1; head -8 gpurun_out/r4_cfg2_kstats_lookahead.txt
