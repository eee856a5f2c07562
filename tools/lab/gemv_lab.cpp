// Skinny-GEMM (decode) lab: sx_gemv on the 13B decoder's weight shapes at M = 16 lock-step sequences, through the C-ABI.
//   tools/lab/gemv_lab [rounds] [split...]    split = sx_gemv_tune(2, split): 1 none, 0 automatic, 2 / 4 / 8 forced split-K factor
// Every variant is compared with variant 0 (max |diff| relative to max |y|) and timed interleaved; GB/s = weight bytes / time.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/seedx_hip.h"

#define HCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
#define SXCHECK(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "sx error %d: %s at line %d\n", r_, sx_last_error(), __LINE__); exit(3); } } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static inline uint32_t rng() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 32); }
static inline float urand() { return (float)(rng() >> 8) * (2.0f / 16777216.0f) - 1.0f; }
static inline uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }

struct Shape { const char* name; int N, K, glu, out32, res; };

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 7;
  std::vector<int> variants;
  for (int i = 2; i < argc; ++i) variants.push_back(atoi(argv[i]));
  if (variants.empty()) variants = {1, 0};
  void* ws = nullptr;
  const size_t ws_bytes = 8u << 20;
  HCHECK(hipMalloc(&ws, ws_bytes));
  HCHECK(hipMemset(ws, 0, ws_bytes));
  const int M = 16;
  const Shape shapes[] = {{"qkv 15360x5120", 15360, 5120, 0, 0, 0}, {"o 5120x5120 +res", 5120, 5120, 0, 1, 1},
                          {"gate|up 27648x5120 glu", 27648, 5120, 1, 0, 0}, {"down 5120x13824 +res", 5120, 13824, 0, 1, 1},
                          {"lm_head 32352x5120", 32352, 5120, 0, 1, 0},
                          {"fixed cost: 5120x256 +res", 5120, 256, 0, 1, 1}, {"fixed cost: 15360x256", 15360, 256, 0, 0, 0}};
  // 40 distinct weight buffers per shape would be the honest cold-cache setting; 6 x 140..330 MB already exceeds the 256 MB
  // MALL, so rotate over NBUF copies
  const int NBUF = 6;
  hipEvent_t e0, e1;
  HCHECK(hipEventCreate(&e0)); HCHECK(hipEventCreate(&e1));
  int bad = 0;
  for (const Shape& s : shapes) {
    const size_t nw = (size_t)s.N * s.K, nx = (size_t)M * s.K, ny = (size_t)M * (s.glu ? s.N / 2 : s.N);
    std::vector<uint16_t> hw(nw), hx(nx);
    for (auto& v : hw) v = f2bf(urand() * 0.05f);
    for (auto& v : hx) v = f2bf(urand());
    std::vector<float> hres(ny);
    for (auto& v : hres) v = urand();
    void* w[NBUF]; void *x, *y, *res;
    for (int b = 0; b < NBUF; ++b) { HCHECK(hipMalloc(&w[b], nw * 2)); HCHECK(hipMemcpy(w[b], hw.data(), nw * 2, hipMemcpyHostToDevice)); }
    HCHECK(hipMalloc(&x, nx * 2)); HCHECK(hipMalloc(&y, ny * 4)); HCHECK(hipMalloc(&res, ny * 4));
    const bool xt = getenv("GEMV_LAB_XTILED") != nullptr;
    if (xt) {  // [K/32][16][32]
      std::vector<uint16_t> t(nx);
      for (int m = 0; m < M; ++m)
        for (int k = 0; k < s.K; ++k) t[((size_t)(k / 32) * 16 + m) * 32 + k % 32] = hx[(size_t)m * s.K + k];
      HCHECK(hipMemcpy(x, t.data(), nx * 2, hipMemcpyHostToDevice));
    } else
    HCHECK(hipMemcpy(x, hx.data(), nx * 2, hipMemcpyHostToDevice));
    HCHECK(hipMemcpy(res, hres.data(), ny * 4, hipMemcpyHostToDevice));
    sx_gemv_args a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.y = y; a.residual = s.res ? (const float*)res : nullptr;
    a.M = M; a.N = s.N; a.K = s.K; a.dtype = SX_BF16; a.out_dtype = s.out32 ? SX_F32 : SX_BF16; a.act = s.glu ? SX_ACT_SILU : SX_ACT_NONE; a.glu = s.glu;
    a.w_layout = 1;
    a.x_layout = xt ? 1 : 0;
    a.workspace = ws; a.workspace_bytes = ws_bytes;
    const size_t ybytes = ny * (s.out32 ? 4 : 2);
    printf("== %-26s M %d\n", s.name, M);
    if (getenv("GEMV_LAB_CLEAR_WS")) HCHECK(hipMemset(ws, 0, ws_bytes));
    std::vector<uint8_t> ref(ybytes), got(ybytes);
    for (size_t vi = 0; vi < variants.size(); ++vi) {
      SXCHECK(sx_gemv_tune(2, variants[vi]));
      a.W = w[0];
      HCHECK(hipMemset(y, 0xff, ybytes));
      SXCHECK(sx_gemv(&a, nullptr));
      HCHECK(hipDeviceSynchronize());
      HCHECK(hipMemcpy(got.data(), y, ybytes, hipMemcpyDeviceToHost));
      {  // repeatability of the split-K result (the order of arrival must not matter): 20 launches
        std::vector<uint8_t> again(ybytes);
        int nrep = 0;
        for (int t = 0; t < 20; ++t) {
          SXCHECK(sx_gemv(&a, nullptr));
          HCHECK(hipDeviceSynchronize());
          HCHECK(hipMemcpy(again.data(), y, ybytes, hipMemcpyDeviceToHost));
          if (memcmp(again.data(), got.data(), ybytes) != 0) nrep++;
        }
        if (nrep) { printf("   split %d: %d of 20 repeated launches differ → BAD\n", variants[vi], nrep); bad++; }
      }
      if (vi == 0) { ref = got; continue; }
      double worst = 0, scale = 0;
      for (size_t i = 0; i < ny; ++i) {
        float r, g;
        if (s.out32) { r = ((float*)ref.data())[i]; g = ((float*)got.data())[i]; }
        else { uint32_t u = (uint32_t)((uint16_t*)ref.data())[i] << 16, v = (uint32_t)((uint16_t*)got.data())[i] << 16; memcpy(&r, &u, 4); memcpy(&g, &v, 4); }
        worst = std::max(worst, (double)fabsf(r - g)); scale = std::max(scale, (double)fabsf(r));
      }
      const bool ok = worst <= 8e-3 * scale && std::isfinite(worst);
      printf("   split %2d vs split %d: max |diff| %.3e (max |y| %.3e)%s → %s\n", variants[vi], variants[0], worst, scale,
             memcmp(ref.data(), got.data(), ybytes) == 0 ? " bit-identical" : "", ok ? "ok" : "BAD");
      if (!ok) bad++;
    }
    const int iters = 60;
    std::vector<std::vector<double>> us(variants.size());
    for (int r = 0; r < rounds + 1; ++r)
      for (size_t vi = 0; vi < variants.size(); ++vi) {
        SXCHECK(sx_gemv_tune(2, variants[vi]));
        HCHECK(hipEventRecord(e0, nullptr));
        for (int it = 0; it < iters; ++it) { a.W = w[it % NBUF]; SXCHECK(sx_gemv(&a, nullptr)); }
        HCHECK(hipEventRecord(e1, nullptr));
        HCHECK(hipEventSynchronize(e1));
        float ms;
        HCHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0) us[vi].push_back(ms * 1e3 / iters);
      }
    for (size_t vi = 0; vi < variants.size(); ++vi) {
      std::sort(us[vi].begin(), us[vi].end());
      const double m = us[vi][us[vi].size() / 2];
      printf("   split %2d: median %7.1f us  %6.0f GB/s (weights)   best %7.1f us\n", variants[vi], m, nw * 2.0 / m * 1e-3, us[vi][0]);
    }
    fflush(stdout);
    for (int b = 0; b < NBUF; ++b) (void)hipFree(w[b]);
    (void)hipFree(x); (void)hipFree(y); (void)hipFree(res);
  }
  SXCHECK(sx_gemv_tune(2, 0));
  printf("%s: %d failing checks\n", bad ? "GEMV LAB FAILED" : "GEMV LAB OK", bad);
  return bad ? 1 : 0;
}
