// GEMM lab: same-process interleaved A/B of sx_gemm tile configs through the C-ABI (no Python / torch start-up on the GPU box).
//   build: tools/lab/build.sh            run (GPU box): tools/lab/gemm_lab [suite] [rounds]
// For every case: (1) correctness of each config against the reference config (bitwise where the arithmetic order is the
// same, max-rel otherwise) plus an fp64 spot check of the reference itself, (2) a race screen (repeat launches must be
// bit-identical), (3) timing: `rounds` interleaved rounds of `iters` launches per config, median TFLOP/s.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/seedx_hip.h"

#define HCHECK(x)                                                                      \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(2);                                                                         \
    }                                                                                  \
  } while (0)
#define SXCHECK(x)                                                              \
  do {                                                                          \
    int r_ = (x);                                                               \
    if (r_ != 0) {                                                              \
      fprintf(stderr, "sx error %d: %s at line %d\n", r_, sx_last_error(), __LINE__); \
      exit(3);                                                                  \
    }                                                                           \
  } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static inline uint32_t rng() {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return (uint32_t)(rng_state >> 32);
}
static inline float urand() { return (float)(rng() >> 8) * (2.0f / 16777216.0f) - 1.0f; }  // [-1, 1)
static inline uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fff + ((u >> 16) & 1);
  return (uint16_t)(u >> 16);
}
static inline float bf2f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

struct Case {
  const char* name;
  int M, N, K;
  int glu, act, res, out32, bias;
  int conv, B, H, W, Cin, stride, ups;  // conv: M = B*Hout*Wout, K = 9*Cin
  int ref_cfg;                          // lock-step reference config
  std::vector<int> cfgs;                // configs to time: 0..8, or 1000 + v = ping-pong 256 variant v, -1 = automatic
};

struct Bufs {
  void *A = nullptr, *W = nullptr, *C = nullptr, *Cref = nullptr;
  float *bias = nullptr, *res = nullptr;
  std::vector<uint16_t> hA, hW;
  size_t c_bytes = 0;
};

static void fill_bf16(std::vector<uint16_t>& h, size_t n, float scale) {
  h.resize(n);
  for (size_t i = 0; i < n; ++i) h[i] = f2bf(urand() * scale);
}

static void set_cfg(int cfg) {
  // 10000 * gm + c: config c with gm tile-rows per traversal group inside an XCD's rectangle (sx_gemm_force_tile(300 + gm); 0 = default 8)
  SXCHECK(sx_gemm_force_tile(300 + cfg / 10000));
  cfg %= 10000;
  SXCHECK(sx_gemm_force_tile(600 + (cfg >= 2000 ? (cfg / 1000 - 1) : 0)));   // 2000 + c: config 1000 + c with tune mask 1 (A/B of epilogue variants)
  if (cfg >= 2000) cfg = 1000 + cfg % 1000;
  if (cfg >= 1000) {
    SXCHECK(sx_gemm_force_tile(400 + (cfg - 1000) % 10));
    SXCHECK(sx_gemm_force_tile(cfg >= 1100 ? 8 : 7));
  } else {
    SXCHECK(sx_gemm_force_tile(400));
    SXCHECK(sx_gemm_force_tile(cfg));
  }
}

static sx_gemm_args make_args(const Case& c, const Bufs& b, void* out) {
  sx_gemm_args a;
  memset(&a, 0, sizeof(a));
  a.A = b.A; a.W = b.W; a.C = out;
  a.bias = c.bias ? b.bias : nullptr;
  a.residual = c.res ? b.res : nullptr;
  a.M = c.M; a.N = c.N; a.K = c.K;
  const int n_out = c.glu ? c.N / 2 : c.N;
  a.ldc = n_out; a.ldr = n_out;
  a.dtype = SX_BF16;
  a.out_dtype = c.out32 ? SX_F32 : SX_BF16;
  a.act = c.act; a.glu = c.glu;
  a.a_mode = c.conv ? SX_A_CONV3X3 : SX_A_LINEAR;
  if (c.conv) {
    a.B = c.B; a.Hin = c.H; a.Win = c.W; a.Cin = c.Cin;
    const int hv = c.ups ? 2 * c.H : c.H, wv = c.ups ? 2 * c.W : c.W;
    a.Hout = (hv + 2 - 3) / c.stride + 1; a.Wout = (wv + 2 - 3) / c.stride + 1;
    a.stride = c.stride; a.upsample = c.ups;
  }
  return a;
}

static double median(std::vector<double> v) {
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

static int run_case(Case c, int rounds, int iters_scale) {
  if (c.conv) {
    const int hv = c.ups ? 2 * c.H : c.H, wv = c.ups ? 2 * c.W : c.W;
    c.M = c.B * ((hv - 1) / c.stride + 1) * ((wv - 1) / c.stride + 1);
    c.K = 9 * c.Cin;
  }
  const int n_out = c.glu ? c.N / 2 : c.N;
  Bufs b;
  const size_t a_elems = c.conv ? (size_t)c.B * c.H * c.W * c.Cin : (size_t)c.M * c.K;
  fill_bf16(b.hA, a_elems, 1.0f);
  fill_bf16(b.hW, (size_t)c.N * c.K, 1.0f / sqrtf((float)c.K));
  HCHECK(hipMalloc(&b.A, a_elems * 2));
  HCHECK(hipMalloc(&b.W, (size_t)c.N * c.K * 2));
  HCHECK(hipMemcpy(b.A, b.hA.data(), a_elems * 2, hipMemcpyHostToDevice));
  HCHECK(hipMemcpy(b.W, b.hW.data(), (size_t)c.N * c.K * 2, hipMemcpyHostToDevice));
  b.c_bytes = (size_t)c.M * n_out * (c.out32 ? 4 : 2);
  HCHECK(hipMalloc(&b.C, b.c_bytes));
  HCHECK(hipMalloc(&b.Cref, b.c_bytes));
  std::vector<float> hbias(c.N), hres;
  for (auto& x : hbias) x = urand();
  HCHECK(hipMalloc(&b.bias, c.N * 4));
  HCHECK(hipMemcpy(b.bias, hbias.data(), c.N * 4, hipMemcpyHostToDevice));
  if (c.res) {
    hres.resize((size_t)c.M * n_out);
    for (auto& x : hres) x = urand() * 4.0f;
    HCHECK(hipMalloc(&b.res, hres.size() * 4));
    HCHECK(hipMemcpy(b.res, hres.data(), hres.size() * 4, hipMemcpyHostToDevice));
  }
  const double flops = 2.0 * c.M * c.N * c.K;
  printf("== %s: M%d N%d K%d glu%d act%d res%d out%s%s\n", c.name, c.M, c.N, c.K, c.glu, c.act, c.res, c.out32 ? "f32" : "bf16",
         c.conv ? " conv" : "");

  // reference
  set_cfg(c.ref_cfg);
  sx_gemm_args ar = make_args(c, b, b.Cref);
  SXCHECK(sx_gemm(&ar, nullptr));
  HCHECK(hipDeviceSynchronize());
  std::vector<uint8_t> href(b.c_bytes), hout(b.c_bytes);
  HCHECK(hipMemcpy(href.data(), b.Cref, b.c_bytes, hipMemcpyDeviceToHost));
  int bad = 0;
  // fp64 spot check of the reference (plain linear epilogues only: guards against a wrong reference)
  if (!c.conv && !c.glu && c.act == 0) {
    double worst = 0;
    for (int t = 0; t < 48; ++t) {
      const int m = rng() % c.M, n = rng() % c.N;
      double s = 0;
      for (int k = 0; k < c.K; ++k) s += (double)bf2f(b.hA[(size_t)m * c.K + k]) * (double)bf2f(b.hW[(size_t)n * c.K + k]);
      if (c.bias) s += hbias[n];
      if (c.res) s += hres[(size_t)m * n_out + n];
      const double got = c.out32 ? ((float*)href.data())[(size_t)m * n_out + n] : bf2f(((uint16_t*)href.data())[(size_t)m * n_out + n]);
      const double err = fabs(got - s) / (fabs(s) + 1.0);
      worst = std::max(worst, err);
    }
    printf("   reference cfg %d vs fp64 spot check: max rel err %.2e %s\n", c.ref_cfg, worst, worst < (c.out32 ? 1e-4 : 1e-2) ? "ok" : "BAD");
    if (!(worst < (c.out32 ? 1e-4 : 1e-2))) bad++;
  }
  // correctness + race screen of every config
  for (int cfg : c.cfgs) {
    if (cfg == c.ref_cfg) continue;
    set_cfg(cfg);
    sx_gemm_args a = make_args(c, b, b.C);
    HCHECK(hipMemset(b.C, 0xff, b.c_bytes));
    SXCHECK(sx_gemm(&a, nullptr));
    HCHECK(hipDeviceSynchronize());
    HCHECK(hipMemcpy(hout.data(), b.C, b.c_bytes, hipMemcpyDeviceToHost));
    size_t ndiff = 0;
    double maxrel = 0;
    const size_t n = (size_t)c.M * n_out;
    for (size_t i = 0; i < n; ++i) {
      const float x = c.out32 ? ((float*)hout.data())[i] : bf2f(((uint16_t*)hout.data())[i]);
      const float r = c.out32 ? ((float*)href.data())[i] : bf2f(((uint16_t*)href.data())[i]);
      if (memcmp(&x, &r, 4) != 0) {
        ndiff++;
        const double e = fabs((double)x - r) / (fabs((double)r) + 1.0);
        if (!(e <= maxrel)) maxrel = e;   // NaN-propagating
      }
    }
    const double tol = c.out32 ? 2e-5 : 8e-3;
    const bool ok = (ndiff == 0) || (maxrel <= tol);
    if (!ok) {   // where are the wrong elements? (first few + a per-256-row-tile / per-16-column histogram of bad entries)
      int shown = 0;
      std::vector<size_t> by_col(n_out / 16 + 1, 0), by_row16(16, 0);
      for (size_t i = 0; i < n; ++i) {
        const float x = c.out32 ? ((float*)hout.data())[i] : bf2f(((uint16_t*)hout.data())[i]);
        const float r = c.out32 ? ((float*)href.data())[i] : bf2f(((uint16_t*)href.data())[i]);
        if (!(fabs((double)x - r) / (fabs((double)r) + 1.0) <= tol)) {
          const size_t m = i / n_out, nn = i % n_out;
          by_col[nn / 16]++; by_row16[(m % 256) / 16]++;
          if (shown++ < 6) printf("      bad [%zu][%zu]: got %g ref %g (diff %g) residual %g residual[m+16] %g [m-16] %g [n+16] %g [n-16] %g\n", m, nn, x, r, x - r,
                                  c.res ? hres[i] : 0.f, c.res && m + 16 < (size_t)c.M ? hres[i + 16 * n_out] : 0.f, c.res && m >= 16 ? hres[i - 16 * n_out] : 0.f,
                                  c.res && nn + 16 < (size_t)n_out ? hres[i + 16] : 0.f, c.res && nn >= 16 ? hres[i - 16] : 0.f);
        }
      }
      printf("      bad per 16-row block of a 256-row tile:");
      for (size_t v : by_row16) printf(" %zu", v);
      printf("\n      bad per 16-column block (first 24):");
      for (size_t q = 0; q < by_col.size() && q < 24; ++q) printf(" %zu", by_col[q]);
      printf("\n");
    }
    // race screen: 20 more launches, each must reproduce the first bit for bit
    size_t races = 0;
    std::vector<uint8_t> h2(b.c_bytes);
    for (int rep = 0; rep < 20; ++rep) {
      SXCHECK(sx_gemm(&a, nullptr));
      if (rep % 5 == 4) {
        HCHECK(hipDeviceSynchronize());
        HCHECK(hipMemcpy(h2.data(), b.C, b.c_bytes, hipMemcpyDeviceToHost));
        if (memcmp(h2.data(), hout.data(), b.c_bytes) != 0) races++;
      }
    }
    printf("   cfg %4d vs ref %d: %zu / %zu differ, max rel %.2e → %s; race screen: %zu mismatching repeats\n", cfg, c.ref_cfg, ndiff, n,
           maxrel, ok ? (ndiff ? "ok (tolerance)" : "bit-identical") : "MISMATCH", races);
    if (!ok || races) bad++;
  }
  // timing
  hipEvent_t e0, e1;
  HCHECK(hipEventCreate(&e0));
  HCHECK(hipEventCreate(&e1));
  const int iters = std::max(3, (int)(iters_scale * 2.0e12 / flops));   // ~2 ms+ per timed burst at 1 PF
  std::vector<std::vector<double>> us(c.cfgs.size());
  for (int r = 0; r < rounds + 1; ++r) {
    for (size_t ci = 0; ci < c.cfgs.size(); ++ci) {
      set_cfg(c.cfgs[ci]);
      sx_gemm_args a = make_args(c, b, b.C);
      HCHECK(hipEventRecord(e0, nullptr));
      for (int it = 0; it < iters; ++it) SXCHECK(sx_gemm(&a, nullptr));
      HCHECK(hipEventRecord(e1, nullptr));
      HCHECK(hipEventSynchronize(e1));
      float ms;
      HCHECK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0) us[ci].push_back(ms * 1e3 / iters);
    }
  }
  for (size_t ci = 0; ci < c.cfgs.size(); ++ci) {
    const double m = median(us[ci]), lo = *std::min_element(us[ci].begin(), us[ci].end());
    printf("   cfg %4d: median %9.1f us  %7.1f TF   (best %9.1f us %7.1f TF)\n", c.cfgs[ci], m, flops / m * 1e-6, lo, flops / lo * 1e-6);
  }
  fflush(stdout);
  set_cfg(-1);
  (void)hipFree(b.A); (void)hipFree(b.W); (void)hipFree(b.C); (void)hipFree(b.Cref); (void)hipFree(b.bias);
  if (b.res) (void)hipFree(b.res);
  return bad;
}

// per-workgroup phase stamps of one config (start, first k-tile landed, main loop done, stores acknowledged)
static void probe_case(Case c, int cfg) {
  Bufs b;
  const size_t a_elems = (size_t)c.M * c.K;
  fill_bf16(b.hA, a_elems, 1.0f);
  fill_bf16(b.hW, (size_t)c.N * c.K, 1.0f / sqrtf((float)c.K));
  const int n_out = c.glu ? c.N / 2 : c.N;
  HCHECK(hipMalloc(&b.A, a_elems * 2));
  HCHECK(hipMalloc(&b.W, (size_t)c.N * c.K * 2));
  HCHECK(hipMemcpy(b.A, b.hA.data(), a_elems * 2, hipMemcpyHostToDevice));
  HCHECK(hipMemcpy(b.W, b.hW.data(), (size_t)c.N * c.K * 2, hipMemcpyHostToDevice));
  b.c_bytes = (size_t)c.M * n_out * (c.out32 ? 4 : 2);
  HCHECK(hipMalloc(&b.C, b.c_bytes));
  HCHECK(hipMalloc(&b.bias, c.N * 4));
  HCHECK(hipMemset(b.bias, 0, c.N * 4));
  if (c.res) { HCHECK(hipMalloc(&b.res, (size_t)c.M * n_out * 4)); HCHECK(hipMemset(b.res, 0, (size_t)c.M * n_out * 4)); }
  const int nblk = 1 << 16;
  unsigned long long* d;
  HCHECK(hipMalloc(&d, (size_t)nblk * 32));
  set_cfg(cfg);
  sx_gemm_args a = make_args(c, b, b.C);
  for (int w = 0; w < 2; ++w) SXCHECK(sx_gemm(&a, nullptr));
  HCHECK(hipMemset(d, 0, (size_t)nblk * 32));
  SXCHECK(sx_gemm_debug_stamps(d));
  hipEvent_t e0, e1;
  HCHECK(hipEventCreate(&e0));
  HCHECK(hipEventCreate(&e1));
  HCHECK(hipEventRecord(e0, nullptr));
  SXCHECK(sx_gemm(&a, nullptr));
  HCHECK(hipEventRecord(e1, nullptr));
  HCHECK(hipDeviceSynchronize());
  float launch_ms = 0;
  HCHECK(hipEventElapsedTime(&launch_ms, e0, e1));
  SXCHECK(sx_gemm_debug_stamps(nullptr));
  std::vector<unsigned long long> h((size_t)nblk * 4);
  HCHECK(hipMemcpy(h.data(), d, (size_t)nblk * 32, hipMemcpyDeviceToHost));
  std::vector<double> pro, mainl, epi;
  for (int i = 0; i < nblk; ++i) {
    const unsigned long long* s = &h[(size_t)i * 4];
    if (!s[3]) continue;
    pro.push_back((double)(s[1] - s[0]));
    mainl.push_back((double)(s[2] - s[1]));
    epi.push_back((double)(s[3] - s[2]));
  }
  if (!pro.empty()) {
    // s_memtime counters of different XCDs are not aligned: only differences inside one workgroup are meaningful. Every CU
    // runs n/256 tiles back to back, so launch time = (n/256) x mean tile time fixes the tick length.
    double tot = 0;
    for (size_t i = 0; i < pro.size(); ++i) tot += pro[i] + mainl[i] + epi[i];
    const double tick = launch_ms * 1e3 / ((double)pro.size() / 256.0 * (tot / pro.size()));
    printf("   probe %-12s cfg %4d: %5zu tiles, launch %7.1f us | per tile (us): prologue %5.2f  main %6.2f  epilogue %6.2f  (tick %.1f MHz)\n",
           c.name, cfg, pro.size(), launch_ms * 1e3, median(pro) * tick, median(mainl) * tick, median(epi) * tick, 1.0 / tick);
  }
  set_cfg(-1);
  (void)hipFree(b.A); (void)hipFree(b.W); (void)hipFree(b.C); (void)hipFree(b.bias); (void)hipFree(d);
  if (b.res) (void)hipFree(b.res);
}

// tile-model sweep: every tile config over the production shape families, JSON lines for tools/fit_tile_model.py
static void model_sweep() {
  struct S { int conv, M, N, K, B, H, Cin; };
  std::vector<S> shapes;
  shapes.push_back({0, 8192, 8192, 2048, 0, 0, 0});
  for (int b : {1, 2, 4, 8, 16}) {
    const int m32 = 2 * b * 1024, m64 = 2 * b * 4096;
    for (auto nk : std::vector<std::pair<int, int>>{{1280, 1280}, {3840, 1280}, {1280, 5120}, {10240, 1280}})
      shapes.push_back({0, m32, nk.first, nk.second, 0, 0, 0});
    for (auto nk : std::vector<std::pair<int, int>>{{640, 640}, {1920, 640}, {640, 2560}, {5120, 640}})
      shapes.push_back({0, m64, nk.first, nk.second, 0, 0, 0});
  }
  for (auto mnk : std::vector<std::array<int, 3>>{{2048, 1664, 1664}, {2048, 4992, 1664}, {2048, 8192, 1664}, {2048, 1664, 8192},
                                                 {165, 15360, 5120}, {165, 5120, 13824}, {520, 5120, 5120}, {1320, 15360, 5120},
                                                 {1040, 15360, 5120}, {1040, 5120, 13824}, {2640, 15360, 5120}, {2640, 5120, 5120},
                                                 {2640, 27648, 5120}, {2640, 5120, 13824}, {32768, 1664, 1664}, {32768, 4992, 1664},
                                                 {32768, 8192, 1664}, {32768, 1664, 8192}, {16384, 16384, 512}})
    shapes.push_back({0, mnk[0], mnk[1], mnk[2], 0, 0, 0});
  for (int b : {1, 2, 4, 8, 16})
    for (auto hc : std::vector<std::array<int, 3>>{{32, 1280, 1280}, {32, 2560, 1280}, {64, 640, 640}, {64, 1280, 640}, {128, 320, 320},
                                                   {128, 640, 320}})
      shapes.push_back({1, 0, hc[2], 0, 2 * b, hc[0], hc[1]});
  for (auto hc : std::vector<std::array<int, 3>>{{128, 512, 512}, {256, 512, 512}, {512, 256, 256}, {1024, 128, 128}, {512, 512, 256}})
    shapes.push_back({1, 0, hc[2], 0, 1, hc[0], hc[1]});   // VAE decoder levels (one image)
  hipEvent_t e0, e1;
  HCHECK(hipEventCreate(&e0));
  HCHECK(hipEventCreate(&e1));
  for (const S& sh : shapes) {
    Case c = {"m", sh.M, sh.N, sh.K, 0, 0, 0, sh.conv ? 1 : 0, 0, sh.conv, sh.B, sh.H, sh.H, sh.Cin, 1, 0, 0, {}};
    if (c.conv) { c.M = c.B * c.H * c.W; c.K = 9 * c.Cin; }
    Bufs b;
    const size_t a_elems = c.conv ? (size_t)c.B * c.H * c.W * c.Cin : (size_t)c.M * c.K;
    HCHECK(hipMalloc(&b.A, a_elems * 2));
    HCHECK(hipMalloc(&b.W, (size_t)c.N * c.K * 2));
    {  // random bf16 bit patterns with sane exponents, generated on the host once per shape
      std::vector<uint16_t> h;
      fill_bf16(h, std::min(a_elems, (size_t)1 << 24), 1.0f);
      for (size_t o = 0; o < a_elems; o += h.size()) HCHECK(hipMemcpy((uint16_t*)b.A + o, h.data(), std::min(h.size(), a_elems - o) * 2, hipMemcpyHostToDevice));
      const size_t wn = (size_t)c.N * c.K;
      fill_bf16(h, std::min(wn, (size_t)1 << 24), 0.03f);
      for (size_t o = 0; o < wn; o += h.size()) HCHECK(hipMemcpy((uint16_t*)b.W + o, h.data(), std::min(h.size(), wn - o) * 2, hipMemcpyHostToDevice));
    }
    b.c_bytes = (size_t)c.M * c.N * (c.out32 ? 4 : 2);
    HCHECK(hipMalloc(&b.C, b.c_bytes));
    const double flops = 2.0 * c.M * c.N * c.K;
    const int iters = std::max(3, (int)(2.0e12 / flops));
    printf("{\"kind\": \"%s\", \"M\": %d, \"N\": %d, \"K\": %d, \"us\": [", c.conv ? "conv" : "linear", c.M, c.N, c.K);
    for (int cfg = 0; cfg <= 8; ++cfg) {
      set_cfg(cfg == 7 ? 1000 : cfg == 8 ? 1100 : cfg);
      sx_gemm_args a = make_args(c, b, b.C);
      double best = 1e30;
      for (int r = 0; r < 3; ++r) {
        HCHECK(hipEventRecord(e0, nullptr));
        for (int it = 0; it < iters; ++it) SXCHECK(sx_gemm(&a, nullptr));
        HCHECK(hipEventRecord(e1, nullptr));
        HCHECK(hipEventSynchronize(e1));
        float ms;
        HCHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0) best = std::min(best, (double)ms * 1e3 / iters);
      }
      printf("%s%.2f", cfg ? ", " : "", best);
    }
    set_cfg(-1);
    printf("]}\n");
    fflush(stdout);
    (void)hipFree(b.A); (void)hipFree(b.W); (void)hipFree(b.C);
  }
}

// LayerNorm-producer launches (sx_gemm_ln: fp32 residual in, fp32 + 16-bit out, per-row sums): the persistent strip kernel
// (forced tile 9) against the ping-pong producer epilogue (strip excluded: 500). C and x16 must agree bit for bit, the row sums
// to fp32 summation noise (and with the fp64 sums of the stored C), repeated strip launches must reproduce EVERYTHING bit for bit.
static int run_ln_case(const char* name, int M, int N, int K, int rounds, int iters_scale) {
  std::vector<uint16_t> hA, hW;
  fill_bf16(hA, (size_t)M * K, 1.0f);
  fill_bf16(hW, (size_t)N * K, 1.0f / sqrtf((float)K));
  void *A, *W, *C[2], *X[2];
  double* S[2];
  float *bias, *res;
  HCHECK(hipMalloc(&A, hA.size() * 2));
  HCHECK(hipMalloc(&W, hW.size() * 2));
  HCHECK(hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
  HCHECK(hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
  const size_t n = (size_t)M * N;
  for (int v = 0; v < 2; ++v) {
    HCHECK(hipMalloc(&C[v], n * 4));
    HCHECK(hipMalloc(&X[v], n * 2));
    HCHECK(hipMalloc((void**)&S[v], (size_t)M * 16));
  }
  std::vector<float> hbias(N), hres(n);
  for (auto& x : hbias) x = urand();
  for (auto& x : hres) x = urand() * 4.0f;
  HCHECK(hipMalloc(&bias, N * 4));
  HCHECK(hipMalloc(&res, n * 4));
  HCHECK(hipMemcpy(bias, hbias.data(), N * 4, hipMemcpyHostToDevice));
  HCHECK(hipMemcpy(res, hres.data(), n * 4, hipMemcpyHostToDevice));
  printf("== %s (LayerNorm producer): M%d N%d K%d\n", name, M, N, K);
  auto launch = [&](int v) {   // v = 0: ping-pong producer, 1: strip kernel
    sx_gemm_args a;
    memset(&a, 0, sizeof(a));
    a.A = A; a.W = W; a.C = C[v]; a.bias = bias; a.residual = res;
    a.M = M; a.N = N; a.K = K; a.ldc = N; a.ldr = N;
    a.dtype = SX_BF16; a.out_dtype = SX_F32; a.a_mode = SX_A_LINEAR;
    sx_gemm_ln_args ln;
    memset(&ln, 0, sizeof(ln));
    ln.x16_out = X[v]; ln.row_stats_out = S[v]; ln.ld_x16 = N;
    SXCHECK(sx_gemm_force_tile(v ? 9 : 8));   // reference: the 256x320 ping-pong producer
    SXCHECK(sx_gemm_force_tile(v ? 501 : 500));
    HCHECK(hipMemsetAsync(S[v], 0, (size_t)M * 16, nullptr));
    SXCHECK(sx_gemm_ln(&a, &ln, nullptr));
  };
  int bad = 0;
  for (int v = 0; v < 2; ++v) {
    HCHECK(hipMemset(C[v], 0xff, n * 4));
    HCHECK(hipMemset(X[v], 0xff, n * 2));
    launch(v);
  }
  HCHECK(hipDeviceSynchronize());
  std::vector<float> c0(n), c1(n);
  std::vector<uint16_t> x0(n), x1(n);
  std::vector<double> s0((size_t)M * 2), s1((size_t)M * 2);
  HCHECK(hipMemcpy(c0.data(), C[0], n * 4, hipMemcpyDeviceToHost));
  HCHECK(hipMemcpy(c1.data(), C[1], n * 4, hipMemcpyDeviceToHost));
  HCHECK(hipMemcpy(x0.data(), X[0], n * 2, hipMemcpyDeviceToHost));
  HCHECK(hipMemcpy(x1.data(), X[1], n * 2, hipMemcpyDeviceToHost));
  HCHECK(hipMemcpy(s0.data(), S[0], (size_t)M * 16, hipMemcpyDeviceToHost));
  HCHECK(hipMemcpy(s1.data(), S[1], (size_t)M * 16, hipMemcpyDeviceToHost));
  // fp64 spot check of the ping-pong reference
  double worst = 0;
  for (int t = 0; t < 32; ++t) {
    const int m = rng() % M, nn = rng() % N;
    double s = hbias[nn] + hres[(size_t)m * N + nn];
    for (int k = 0; k < K; ++k) s += (double)bf2f(hA[(size_t)m * K + k]) * (double)bf2f(hW[(size_t)nn * K + k]);
    worst = std::max(worst, fabs(c0[(size_t)m * N + nn] - s) / (fabs(s) + 1.0));
  }
  printf("   ping-pong producer vs fp64 spot check: max rel err %.2e %s\n", worst, worst < 1e-4 ? "ok" : "BAD");
  if (!(worst < 1e-4)) bad++;
  size_t dc = 0, dx = 0;
  int shown = 0;
  for (size_t i = 0; i < n; ++i) {
    if (memcmp(&c0[i], &c1[i], 4) != 0) {
      dc++;
      if (shown++ < 8) printf("      C differs at [%zu][%zu]: strip %g ping-pong %g\n", i / N, i % N, c1[i], c0[i]);
    }
    if (x0[i] != x1[i]) {
      if (dx < 24 || (dx % 97) == 0) printf("      x16 differs at [%zu][%zu]: strip %04x ping-pong %04x (C %g)\n", i / N, i % N, x1[i], x0[i], c1[i]);
      dx++;
    }
  }
  double ws = 0, wsum64 = 0;
  for (int m = 0; m < M; ++m) {
    double e1 = 0, e2 = 0;
    for (int c = 0; c < N; ++c) { const double v = c1[(size_t)m * N + c]; e1 += v; e2 += v * v; }
    for (int q = 0; q < 2; ++q) {
      const double ref = q ? e2 : e1;
      wsum64 = std::max(wsum64, fabs(s1[2 * m + q] - ref) / (fabs(ref) + (q ? e2 : sqrt(e2 * N)) * 1e-1 + 1e-30));
      ws = std::max(ws, fabs(s1[2 * m + q] - s0[2 * m + q]) / (fabs(s0[2 * m + q]) + (q ? e2 : sqrt(e2 * N)) * 1e-1 + 1e-30));
    }
  }
  const bool ok = dc == 0 && dx == 0 && ws < 2e-5 && wsum64 < 2e-5;
  printf("   strip vs ping-pong: C %zu / %zu differ, x16 %zu differ, row sums max rel diff %.2e (vs fp64 sums of the stored C %.2e) → %s\n", dc, n, dx,
         ws, wsum64, ok ? "ok" : "MISMATCH");
  if (!ok) bad++;
  // reproducibility of the strip kernel: C, x16 AND the row sums bit for bit
  size_t races = 0;
  std::vector<float> c2(n);
  std::vector<uint16_t> x2(n);
  std::vector<double> s2((size_t)M * 2);
  for (int rep = 0; rep < 12; ++rep) {
    launch(1);
    if (rep % 4 == 3) {
      HCHECK(hipDeviceSynchronize());
      HCHECK(hipMemcpy(c2.data(), C[1], n * 4, hipMemcpyDeviceToHost));
      HCHECK(hipMemcpy(x2.data(), X[1], n * 2, hipMemcpyDeviceToHost));
      HCHECK(hipMemcpy(s2.data(), S[1], (size_t)M * 16, hipMemcpyDeviceToHost));
      if (memcmp(c2.data(), c1.data(), n * 4) || memcmp(x2.data(), x1.data(), n * 2) || memcmp(s2.data(), s1.data(), (size_t)M * 16)) races++;
    }
  }
  printf("   strip kernel repeat screen: %zu mismatching repeats\n", races);
  if (races) bad++;
  // timing
  hipEvent_t e0, e1;
  HCHECK(hipEventCreate(&e0));
  HCHECK(hipEventCreate(&e1));
  const double flops = 2.0 * M * N * K, bytes = (double)n * (4 + 4 + 2) + (double)M * K * 2 + (double)N * K * 2;
  const int iters = std::max(3, (int)(iters_scale * 2.0e12 / flops));
  std::vector<double> us[2];
  for (int r = 0; r < rounds + 1; ++r)
    for (int v = 0; v < 2; ++v) {
      HCHECK(hipEventRecord(e0, nullptr));
      for (int it = 0; it < iters; ++it) launch(v);
      HCHECK(hipEventRecord(e1, nullptr));
      HCHECK(hipEventSynchronize(e1));
      float ms;
      HCHECK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0) us[v].push_back(ms * 1e3 / iters);
    }
  for (int v = 0; v < 2; ++v) {
    const double m = median(us[v]);
    printf("   %-10s median %9.1f us  %7.1f TF  %6.2f TB/s (incl. the statistics memset)\n", v ? "strip" : "ping-pong", m, flops / m * 1e-6, bytes / m * 1e-6);
  }
  fflush(stdout);
  SXCHECK(sx_gemm_force_tile(-1));
  SXCHECK(sx_gemm_force_tile(501));
  (void)hipFree(A); (void)hipFree(W); (void)hipFree(bias); (void)hipFree(res);
  for (int v = 0; v < 2; ++v) { (void)hipFree(C[v]); (void)hipFree(X[v]); (void)hipFree(S[v]); }
  return bad;
}

// s_memtime spans of the strip kernel's rolled phases (PROBE build, sx_gemm_debug_stamps)
static void probe_strip(int M, int N, int K, int abl = 0) {
  void *A, *W, *C, *X;
  double* S;
  float *bias, *res;
  const size_t n = (size_t)M * N;
  HCHECK(hipMalloc(&A, (size_t)M * K * 2)); HCHECK(hipMemset(A, 0x11, (size_t)M * K * 2));
  HCHECK(hipMalloc(&W, (size_t)N * K * 2)); HCHECK(hipMemset(W, 0x11, (size_t)N * K * 2));
  HCHECK(hipMalloc(&C, n * 4)); HCHECK(hipMalloc(&X, n * 2)); HCHECK(hipMalloc((void**)&S, (size_t)M * 16));
  HCHECK(hipMalloc(&bias, N * 4)); HCHECK(hipMemset(bias, 0, N * 4));
  HCHECK(hipMalloc(&res, n * 4)); HCHECK(hipMemset(res, 0, n * 4));
  unsigned long long* d;
  const size_t dn = 256 * 8 * 8;
  HCHECK(hipMalloc(&d, dn * 8));
  HCHECK(hipMemset(d, 0, dn * 8));
  sx_gemm_args a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.W = W; a.C = C; a.bias = bias; a.residual = res;
  a.M = M; a.N = N; a.K = K; a.ldc = N; a.ldr = N;
  a.dtype = SX_BF16; a.out_dtype = SX_F32; a.a_mode = SX_A_LINEAR;
  sx_gemm_ln_args ln;
  memset(&ln, 0, sizeof(ln));
  ln.x16_out = X; ln.row_stats_out = S; ln.ld_x16 = N;
  SXCHECK(sx_gemm_force_tile(9));
  SXCHECK(sx_gemm_force_tile(501));
  SXCHECK(sx_gemm_force_tile(300 + abl));
  SXCHECK(sx_gemm_debug_stamps(d));
  hipEvent_t e0, e1;
  HCHECK(hipEventCreate(&e0)); HCHECK(hipEventCreate(&e1));
  SXCHECK(sx_gemm_ln(&a, &ln, nullptr));
  HCHECK(hipEventRecord(e0, nullptr));
  for (int w = 0; w < 5; ++w) SXCHECK(sx_gemm_ln(&a, &ln, nullptr));
  HCHECK(hipEventRecord(e1, nullptr));
  HCHECK(hipDeviceSynchronize());
  float ms; HCHECK(hipEventElapsedTime(&ms, e0, e1));
  SXCHECK(sx_gemm_debug_stamps(nullptr));
  SXCHECK(sx_gemm_force_tile(300));
  printf("   ablation mask %d: %.1f us per launch (probe build)\n", abl, ms * 200.0);
  std::vector<unsigned long long> h(dn);
  HCHECK(hipMemcpy(h.data(), d, dn * 8, hipMemcpyDeviceToHost));
  const char* nm[6] = {"issue(reads+dma)", "vmcnt", "lgkmcnt", "barrier1", "mfma", "barrier2"};
  printf("== strip probe M%d N%d K%d (s_memtime ticks per rolled phase, mean over the workgroups; group 0 = waves 0-3, group 1 = waves 4-7)\n", M, N, K);
  const int nb = std::min(256, M / 128);
  for (int grp = 0; grp < 2; ++grp) {
    double acc[8] = {0};
    int cnt = 0;
    for (int b = 0; b < nb; ++b)
      for (int w = grp * 4; w < grp * 4 + 4; ++w) {
        const unsigned long long* q = &h[((size_t)b * 8 + w) * 8];
        if (!q[6]) continue;
        for (int k = 0; k < 6; ++k) acc[k] += (double)q[k] / (double)q[6];
        acc[7] += (double)q[7];
        cnt++;
      }
    if (!cnt) continue;
    printf("   group %d:", grp);
    double tot = 0;
    for (int k = 0; k < 6; ++k) { printf(" %s %.0f", nm[k], acc[k] / cnt); tot += acc[k] / cnt; }
    printf(" | per phase %.0f ticks, kernel %.0f ticks\n", tot, acc[7] / cnt);
  }
  SXCHECK(sx_gemm_force_tile(-1));
  (void)hipFree(A); (void)hipFree(W); (void)hipFree(C); (void)hipFree(X); (void)hipFree(S); (void)hipFree(bias); (void)hipFree(res); (void)hipFree(d);
}

int main(int argc, char** argv) {
  const std::string suite = argc > 1 ? argv[1] : "core";
  const int rounds = argc > 2 ? atoi(argv[2]) : 5;
  const int scale = argc > 3 ? atoi(argv[3]) : 4;
  hipDeviceProp_t prop;
  HCHECK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s, %d CUs, clock %d MHz; sx_version %d\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000, sx_version());
  if (suite == "model") { model_sweep(); return 0; }
  if (suite == "glu") {    // GLU epilogue: 16-byte stores (1000) against the 8-byte stores (2000), same box, interleaved
    int bad = 0;
    bad += run_case({"geglu", 32768, 10240, 1280, 1, 1, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 4, {1000, 2000}}, rounds, scale);
    bad += run_case({"c640_geglu", 131072, 5120, 640, 1, 1, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 4, {1000, 2000}}, rounds, scale);
    bad += run_case({"llm_gateup", 2640, 27648, 5120, 1, 2, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 4, {1000, 2000}}, rounds, scale);
    printf("%s: %d failing checks\n", bad ? "LAB FAILED" : "LAB OK", bad);
    return bad ? 1 : 0;
  }
  if (suite == "gm") {     // tile traversal: which operand panel stays L2-resident across an XCD's consecutive rounds (raster part)
    int bad = 0;
    bad += run_case({"geglu", 32768, 10240, 1280, 1, 1, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 4, {1000, 41000, 21000, 161000, 321000}}, rounds, scale);
    bad += run_case({"qkv", 32768, 3840, 1280, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 5, {1100, 41100, 21100, 161100}}, rounds, scale);
    bad += run_case({"ff2_res", 32768, 1280, 5120, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 1, 0, 5, {1100, 41100, 21100, 161100}}, rounds, scale);
    bad += run_case({"c640_geglu", 131072, 5120, 640, 1, 1, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 4, {1000, 41000, 21000, 161000}}, rounds, scale);
    printf("%s: %d failing checks\n", bad ? "LAB FAILED" : "LAB OK", bad);
    return bad ? 1 : 0;
  }
  if (suite == "stripprobe") {
    for (int abl : {0, 1, 4, 5, 2, 8})
      for (int m : {1024, 32768}) probe_strip(m, 1280, 5120, abl);
    return 0;
  }
  if (suite == "strip" || suite == "stripq") {   // LayerNorm producers: the persistent strip kernel against the ping-pong producer epilogue
    int bad = 0;
    bad += run_ln_case("strip_small", 1024, 1280, 1280, rounds, scale);
    bad += run_ln_case("strip_3strips_k704", 384, 256, 704, rounds, scale);
    bad += run_ln_case("strip_uneven", 128 * 300, 512, 704, rounds, scale);
    bad += run_ln_case("outproj_res_ln", 32768, 1280, 1280, rounds, scale);
    if (suite == "strip") {
      bad += run_ln_case("ff2_res_ln", 32768, 1280, 5120, rounds, scale);
      bad += run_ln_case("c1280_batch8", 16384, 1280, 1280, rounds, scale);
      bad += run_ln_case("n2560", 32768, 2560, 1280, rounds, scale);
    }
    printf("%s: %d failing checks\n", bad ? "LAB FAILED" : "LAB OK", bad);
    return bad ? 1 : 0;
  }
  std::vector<Case> cases;
  // name, M, N, K, glu, act, res, out32, bias, conv, B, H, W, Cin, stride, ups, ref, cfgs
  if (suite == "core" || suite == "all") {
    cases.push_back({"sq8192", 8192, 8192, 8192, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 4, {4, 1000, 1001, 1002, 1003, 5, 1100}});
    cases.push_back({"geglu", 32768, 10240, 1280, 1, 1, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 4, {4, 1000}});
    cases.push_back({"qkv", 32768, 3840, 1280, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 5, {5, 1100, 4, 1000, 1001, 1002, 1003}});
    cases.push_back({"outproj_res", 32768, 1280, 1280, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 1, 0, 5, {5, 1100, 4, 1000}});
    cases.push_back({"ff2_res", 32768, 1280, 5120, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 1, 0, 5, {5, 1100, 1000}});
    cases.push_back({"q_16b", 32768, 1280, 1280, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 5, {5, 1100, 1000}});
    cases.push_back({"c640_out_res", 131072, 640, 640, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 1, 0, 5, {5, 1100, 4, 1000}});
    cases.push_back({"c640_geglu", 131072, 5120, 640, 1, 1, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 4, {4, 1000}});
    cases.push_back({"vit_fc", 32768, 8192, 1664, 0, 1, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 4, {4, 1000}});
    cases.push_back({"conv1280", 0, 1280, 0, 0, 0, 1, 1, 1, 1, 32, 32, 32, 1280, 1, 0, 5, {5, 1100, 1000}});
    cases.push_back({"conv320", 0, 320, 0, 0, 0, 0, 1, 1, 1, 32, 128, 128, 320, 1, 0, 5, {5, 1100}});
    cases.push_back({"conv_up640", 0, 640, 0, 0, 0, 0, 1, 1, 1, 32, 32, 32, 640, 1, 1, 5, {5, 1100}});
    cases.push_back({"conv_s2_640", 0, 640, 0, 0, 0, 0, 1, 1, 1, 32, 64, 64, 640, 2, 0, 5, {5, 1100}});
  }
  if (suite == "hbm") {   // the fp32-residual shapes whose roofline is HBM: every tile config of the menu
    cases.push_back({"outproj_res", 32768, 1280, 1280, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 1, 0, 5, {5, 0, 1, 2, 3, 4, 6, 7, 8}});
    cases.push_back({"c640_out_res", 131072, 640, 640, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 1, 0, 5, {5, 0, 1, 2, 3, 4, 6, 7, 8}});
    cases.push_back({"c320_out_res", 524288, 320, 320, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 1, 0, 5, {5, 0, 1, 2, 3, 4, 6, 7, 8}});
  }
  if (suite == "io") {    // K = 64: one k-tile, the launch is the residual pre-load + the store epilogue (336 MB of fp32 in + out)
    cases.push_back({"res_k64", 32768, 1280, 64, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 1, 0, 5, {5, 8, 7, 0}});
    cases.push_back({"nores_k64", 32768, 1280, 64, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 1, 0, 5, {5, 8, 7, 0}});
    cases.push_back({"bf16_k64", 32768, 1280, 64, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 5, {5, 8, 7, 0}});
  }
  if (suite == "rs") {
    cases.push_back({"rs_small", 1024, 1280, 1280, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 1, 0, 5, {5, 1100, 1000}});
    cases.push_back({"outproj_res", 32768, 1280, 1280, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 1, 0, 5, {5, 1100, 1000}});
    cases.push_back({"rs_k1536", 4096, 1280, 1536, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 1, 0, 5, {5, 1100, 1000}});
    cases.push_back({"ff2_res", 32768, 1280, 5120, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 1, 0, 5, {5, 1100, 1000}});
  }
  if (suite == "edge" || suite == "all") {
    // ragged M / N, K = 64 (one k-tile), K = 128, odd k-tile counts, n_valid-free GLU
    cases.push_back({"edge_k64", 1000, 1280, 64, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 5, {5, 1100, 1000}});
    cases.push_back({"edge_k128", 777, 3840, 128, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 1, 0, 5, {5, 1100, 1000}});
    cases.push_back({"edge_k192", 2640, 5120, 192, 0, 1, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 4, {4, 1000, 1100}});
    cases.push_back({"edge_glu", 2640, 27648, 320, 1, 2, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 4, {4, 1000}});
    cases.push_back({"edge_n", 4100, 1296, 1280, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 1, 0, 5, {5, 1100, 1000}});
  }
  int bad = 0;
  for (auto& c : cases) bad += run_case(c, rounds, scale);
  if (suite == "core" || suite == "all" || suite == "probe") {
    Case g = {"geglu", 32768, 10240, 1280, 1, 1, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 4, {}};
    probe_case(g, 4); probe_case(g, 1000);
    Case q = {"qkv", 32768, 3840, 1280, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 5, {}};
    probe_case(q, 5); probe_case(q, 1100); probe_case(q, 1000);
    Case o = {"outproj_res", 32768, 1280, 1280, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 1, 0, 5, {}};
    probe_case(o, 5); probe_case(o, 1100);
    Case f = {"ff2_res", 32768, 1280, 5120, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 1, 0, 5, {}};
    probe_case(f, 5); probe_case(f, 1100);
  }
  printf("%s: %d failing checks\n", bad ? "LAB FAILED" : "LAB OK", bad);
  return bad ? 1 : 0;
}
