// Probe: element routing of ds_read_b64_tr_b16 on gfx950 (one wave). LDS is filled with lds[i] = i (16-bit), every lane
// reads 8 bytes at byte address 8*lane (+ optional stride scheme) and the 4 returned 16-bit values per lane are printed.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void probe(unsigned short* out, int mode) {
  __shared__ unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int lane = threadIdx.x;
  unsigned addr;
  if (mode == 0) addr = lane * 8;                       // 64 consecutive 8-byte chunks
  else if (mode == 1) addr = (lane & 15) * 64 + (lane >> 4) * 8;   // 16 rows of 64 B, lane group g reads chunk g of each row
  else addr = (lane & 15) * 128 + (lane >> 4) * 8;
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)((__attribute__((address_space(3))) char*)lds + addr));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (unsigned short)r[j];
}
int main() {
  unsigned short* d;
  hipMalloc(&d, 64 * 4 * 2);
  unsigned short h[256];
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  }
  return 0;
}
