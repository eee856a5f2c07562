#!/bin/bash
# Build libseedx_hip.so for gfx950 (cross-compiles without a GPU). Usage: build.sh [--force] [outdir]
#   --force: recompile every translation unit (what __graft_entry__.build() does); default: only TUs older than their sources
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
FORCE=0
if [ "$1" = "--force" ]; then FORCE=1; shift; fi
OUT="${1:-$HERE/../lib}"
mkdir -p "$OUT" "$HERE/.obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result"
FAST="-ffast-math -fno-finite-math-only"
if [ "$FORCE" = 1 ]; then rm -f "$HERE"/.obj/*.o; fi
pids=()
for f in gemm gemm_pp gemm_strip norm attn elementwise decode preproc comm precise; do
  src="$HERE/$f.hip"; obj="$HERE/.obj/$f.o"
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ "$HERE/sx_common.h" -nt "$obj" ] || [ "$HERE/gemm_common.h" -nt "$obj" ] || [ "$HERE/../../include/seedx_hip.h" -nt "$obj" ]; then
    extra="$FAST"
    if [ "$f" = "preproc" ] || [ "$f" = "precise" ]; then extra=""; fi   # integer / IEEE-exact float work, hi + lo operand splits: no fast-math
    # MFMA accumulators in arch VGPRs: the softmax / rescale VALU code touches every accumulator each KV tile, the
    # default AGPR form costs ~200 v_accvgpr_read/write per tile (attention only; the GEMM touches them once)
    if [ "$f" = "attn" ] || [ -n "$SX_VGPR_FORM_ALL" ]; then extra="$extra -mllvm -amdgpu-mfma-vgpr-form=1"; fi
    $HIPCC $BASE $extra -c "$src" -o "$obj" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libseedx_hip.so" "$HERE"/.obj/*.o
echo "built $OUT/libseedx_hip.so"
