// GroupNorm statistics / apply lab: sx_groupnorm_sp phase 1 (statistics) and phase 2 (apply) timed separately on the UNet shapes.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../include/seedx_hip.h"
#define HCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
#define SXCHECK(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "sx error: %s line %d\n", sx_last_error(), __LINE__); exit(3); } } while (0)
int main(int argc, char** argv) {
  struct S { int B, HW, C, C1; };
  const S shapes[] = {{16, 1024, 1280, 0}, {16, 4096, 640, 0}, {16, 16384, 320, 0}, {16, 1024, 2560, 1280}, {16, 4096, 1920, 1280}, {16, 16384, 960, 640},
                      {1, 1048576, 128, 0}, {1, 262144, 256, 0}};
  std::vector<int> blocks = {0, 512, 1024, 2048, 4096};
  hipEvent_t e0, e1;
  HCHECK(hipEventCreate(&e0)); HCHECK(hipEventCreate(&e1));
  for (const S& s : shapes) {
    const size_t n = (size_t)s.B * s.HW * s.C;
    float *x, *gamma, *beta; void* y; double* st;
    HCHECK(hipMalloc(&x, n * 4)); HCHECK(hipMalloc(&y, n * 2)); HCHECK(hipMalloc(&gamma, s.C * 4)); HCHECK(hipMalloc(&beta, s.C * 4));
    HCHECK(hipMalloc(&st, s.B * 32 * 2 * 8));
    HCHECK(hipMemset(x, 0x3c, n * 4)); HCHECK(hipMemset(gamma, 0, s.C * 4)); HCHECK(hipMemset(beta, 0, s.C * 4));
    const float* x2 = s.C1 ? x + (size_t)s.B * s.HW * s.C1 : nullptr;
    printf("B%d HW%d C%d%s (%.0f MB fp32):", s.B, s.HW, s.C, s.C1 ? " two-source" : "", n * 4 / 1e6);
    for (int nb : blocks) {
      SXCHECK(sx_norm_tune(0, nb));
      std::vector<float> us;
      for (int r = 0; r < 7; ++r) {
        HCHECK(hipEventRecord(e0, nullptr));
        for (int it = 0; it < 5; ++it) SXCHECK(sx_groupnorm_sp(x, x2, s.C1, y, nullptr, SX_BF16, gamma, beta, st, s.B, s.HW, s.HW, s.C, 32, 1e-5f, 1, 1, nullptr));
        HCHECK(hipEventRecord(e1, nullptr)); HCHECK(hipEventSynchronize(e1));
        float ms; HCHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2) us.push_back(ms * 1e3f / 5);
      }
      std::sort(us.begin(), us.end());
      printf("  stats@%d %.1f us (%.2f TB/s)", nb, us[us.size() / 2], n * 4 / us[us.size() / 2] * 1e-6);
    }
    {
      std::vector<float> us;
      for (int r = 0; r < 7; ++r) {
        HCHECK(hipEventRecord(e0, nullptr));
        for (int it = 0; it < 5; ++it) SXCHECK(sx_groupnorm_sp(x, x2, s.C1, y, nullptr, SX_BF16, gamma, beta, st, s.B, s.HW, s.HW, s.C, 32, 1e-5f, 1, 2, nullptr));
        HCHECK(hipEventRecord(e1, nullptr)); HCHECK(hipEventSynchronize(e1));
        float ms; HCHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2) us.push_back(ms * 1e3f / 5);
      }
      std::sort(us.begin(), us.end());
      printf("  | apply %.1f us (%.2f TB/s r+w)\n", us[us.size() / 2], n * 6 / us[us.size() / 2] * 1e-6);
    }
    (void)hipFree(x); (void)hipFree(y); (void)hipFree(gamma); (void)hipFree(beta); (void)hipFree(st);
  }
  SXCHECK(sx_norm_tune(0, 0));
  return 0;
}
