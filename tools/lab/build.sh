#!/bin/bash
# Build the C++ lab harnesses against the in-tree libseedx_hip.so (host code only; run on the GPU box).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
LIB="$HERE/../../seed-x_amd/lib"
for t in gemm_lab attn_lab gn_lab gemv_lab; do
  g++ -O2 -std=c++17 -Wno-unused-result -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -o "$HERE/$t" "$HERE/$t.cpp" \
      -L"$LIB" -lseedx_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$ORIGIN/../../seed-x_amd/lib' -Wl,-rpath,/opt/rocm/lib
done
echo "built lab tools in $HERE"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result "$HERE/tile_io_bench.hip" -o "$HERE/tile_io_bench"
