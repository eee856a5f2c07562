#!/bin/bash
# FIRST CONTACT WITH AN 8-GPU NODE, one command. Nothing in this repository has ever crossed xGMI: no multi-GPU
# box was available to any build round (SCALE_r01..r05 are skipped records). Run from the repo root on a node with N >= 2 MI355X:
#
#     bash tools/runs/r6_scale.sh [outdir]            (default gpurun_out/r6_scale)
#
# Produces, in this order (each step is independent; a failing one is reported and skipped):
#   1. scale_N{1,2,4,8}.json      bench.py's headline line at 1 / 2 / 4 / 8 replicas (throughput mode: independent generations, no
#                                 data-path collective; `ranks_seen` in the line proves which devices the ranks bound)
#   2. tp_config{0,4}_N{4,8}.json latency mode of ONE generation over N GPUs (Megatron-TP Llama with the one-shot IPC all-reduce
#                                 inside the decode graph + pixel-row-sharded UNet), t2i (config 0 path) and edit (BASELINE config 4)
#   3. tp_step_trace_N8/          rocprofv3 kernel trace of a TP = 8 generation: does the K|V all-gather of every self-attention
#                                 (side stream) really sit under the Q projection, and the halo exchange under the conv? DESIGN.md §6's
#                                 model says ≈ 0.8 ms exposed per UNet forward; the trace is what validates or refutes it
#   4. ipc_selftest.log           IpcComm's start-up self-test verdicts of every rank (a failing rank routes its collectives to RCCL
#                                 and says so here)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=${1:-$R/gpurun_out/r6_scale}
mkdir -p "$O"
cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
NG=$(python - <<'PY'
import torch
print(torch.cuda.device_count())
PY
)
echo "visible GPUs: $NG" | tee "$O/devices.txt"
rocm-smi --showtopo >> "$O/devices.txt" 2>&1 || true
for n in 1 2 4 8; do
  [ "$n" -le "$NG" ] || continue
  echo "== bench.py --gpus $n"
  timeout 1500 python bench.py --gpus $n --steps 5 --warmup 2 $([ $n -gt 1 ] && echo --no-roofline --no-cpu-baseline --also-dtype none) \
      > "$O/scale_N$n.json" 2> "$O/scale_N$n.err" || echo "   bench --gpus $n failed (rc $?): see $O/scale_N$n.err"
  python - "$O/scale_N$n.json" <<'PY' || true
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("   value %.3f gens/s on %d GPU(s); ranks_seen devices: %s" % (r["value"], r["n_gpus"], sorted({g.get("device_id") for g in r["ranks_seen"]})))
PY
done
for cfg in 0 4; do
  for n in 4 8; do
    [ "$n" -le "$NG" ] || continue
    echo "== bench_tp_latency.py --gpus $n --config $cfg (ipc collectives)"
    timeout 1500 python tools/bench_tp_latency.py --gpus $n --config $cfg --steps 3 --warmup 1 > "$O/tp_config${cfg}_N$n.json" 2> "$O/tp_config${cfg}_N$n.err" \
        || echo "   failed (rc $?): see $O/tp_config${cfg}_N$n.err"
    grep -h "self-test\|fall" "$O/tp_config${cfg}_N$n.err" >> "$O/ipc_selftest.log" 2>/dev/null || true
    echo "== same, RCCL collectives (the fallback path, for the A/B)"
    timeout 1500 python tools/bench_tp_latency.py --gpus $n --config $cfg --steps 3 --warmup 1 --llm-comm rccl --unet-comm rccl \
        > "$O/tp_config${cfg}_N${n}_rccl.json" 2> "$O/tp_config${cfg}_N${n}_rccl.err" || echo "   failed (rc $?)"
  done
done
if [ "$NG" -ge 8 ]; then
  echo "== rocprofv3 kernel trace of one TP = 8 generation"
  ( cd /tmp && export TMPDIR=/tmp && timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/tp_step_trace_N8" -- \
      python "$R/tools/bench_tp_latency.py" --gpus 8 --config 0 --steps 1 --warmup 1 > "$O/tp_step_trace_N8.log" 2>&1 ) || echo "   trace failed"
fi
echo "done: $O  — copy scale_N*.json / tp_config*.json into profiles/ and replace DESIGN.md §6's 'not measured' statements with the numbers"
