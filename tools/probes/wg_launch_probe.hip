// Probe: cost of one "round" of 256 workgroups (512 threads, 128 KiB dynamic LDS = one block per CU) with an empty body,
// i.e. the per-tile workgroup launch / drain overhead a non-persistent 256x256 GEMM pays per round of tiles.
#include <hip/hip_runtime.h>
#include <stdio.h>
extern __shared__ unsigned char smem[];
__global__ __launch_bounds__(512) void empty_kernel(int* sink, int spin) {
  if (spin) {                       // optional fixed work: `spin` dependent LDS round trips per wave
    volatile unsigned* s = (volatile unsigned*)smem;
    unsigned v = threadIdx.x;
    for (int i = 0; i < spin; ++i) { s[threadIdx.x] = v; v = s[threadIdx.x ^ 1] + 1; }
    if (v == 0xdeadbeef) *sink = 1;
  }
}
int main() {
  int* d;
  hipMalloc(&d, 4);
  hipFuncSetAttribute((const void*)empty_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int lds : {131072, 65536, 16384}) {
    for (int spin : {0, 200}) {
      for (int rounds : {1, 10, 40}) {
        const int grid = 256 * rounds;
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(512), lds, 0, d, spin);
        hipEventRecord(a);
        const int it = 50;
        for (int i = 0; i < it; ++i) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(512), lds, 0, d, spin);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        printf("lds %6d spin %3d rounds %2d: %8.2f us per launch, %6.2f us per round\n", lds, spin, rounds, ms * 1e3 / it,
               ms * 1e3 / it / rounds);
      }
    }
  }
  return 0;
}
