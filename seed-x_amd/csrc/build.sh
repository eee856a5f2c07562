#!/bin/bash
# Build libseedx_hip.so for gfx950 (cross-compiles without a GPU). Usage: build.sh [outdir]
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="${1:-$HERE/../lib}"
mkdir -p "$OUT" "$HERE/.obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -Wno-unused-result"
pids=()
for f in gemm norm attn elementwise decode; do
  src="$HERE/$f.hip"; obj="$HERE/.obj/$f.o"
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ "$HERE/sx_common.h" -nt "$obj" ] || [ "$HERE/../../include/seedx_hip.h" -nt "$obj" ]; then
    extra=""
    # MFMA accumulators in arch VGPRs: the softmax / rescale VALU code touches every accumulator each KV tile, the
    # default AGPR form costs ~200 v_accvgpr_read/write per tile (attention only; the GEMM touches them once)
    if [ "$f" = "attn" ] || [ -n "$SX_VGPR_FORM_ALL" ]; then extra="-mllvm -amdgpu-mfma-vgpr-form=1"; fi
    $HIPCC $FLAGS $extra -c "$src" -o "$obj" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libseedx_hip.so" "$HERE"/.obj/*.o
echo "built $OUT/libseedx_hip.so"
