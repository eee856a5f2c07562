// How fast can 256 workgroups (8 waves) each read / write one 256 x BN fp32 or bf16 tile of a row-major [M][N] matrix?
// Compares the MFMA-fragment access pattern of the GEMM epilogue (a wave instruction = 16 rows x 64 B) with row-contiguous
// patterns (a wave instruction = 1 KiB of one row, or 4 rows x 256 B). Answers whether the residual read / C store of the
// K = 1280 GEMMs is bound by the pattern or by HBM.   hipcc --offload-arch=gfx950 -O3 tile_io_bench.hip -o tile_io_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef __attribute__((ext_vector_type(4))) float f32x4_t;
#define HCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

// MODE 0: fragment pattern (lane: row = base + (lane & 15), 16 B at column 4*(lane >> 4) + 16*i), wave (g, wc) as in gemm_pp
// MODE 1: row pattern: wave w covers rows w*32 .. w*32+31; an instruction reads lane*16 B of the row (BN*4 bytes per row)
// WRITE 0: read and reduce; 1: write constants; 2: read + write (copy into out)
template <int BN, int MODE, int WRITE>
__global__ __launch_bounds__(512) void tile_io(const float* __restrict__ in, float* __restrict__ out, int ld, int tiles_n, float* sink) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const size_t base = (size_t)tm * 256 * ld + (size_t)tn * BN;
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  if (MODE == 0) {
    constexpr int TN = BN / 4, FN = TN / 16;
    const int g = wave >> 2, wc = wave & 3;
    f32x4_t v[8][FN];
    if (WRITE != 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < FN; ++i)
          v[j][i] = *(const f32x4_t*)(in + base + (size_t)(g * 128 + j * 16 + (lane & 15)) * ld + wc * TN + i * 16 + (lane >> 4) * 4);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int i = 0; i < FN; ++i) {
        if (WRITE == 0) acc += v[j][i];
        else {
          const f32x4_t o = WRITE == 1 ? (f32x4_t){1.f, 2.f, 3.f, (float)lane} : v[j][i];
          *(f32x4_t*)(out + base + (size_t)(g * 128 + j * 16 + (lane & 15)) * ld + wc * TN + i * 16 + (lane >> 4) * 4) = o;
        }
      }
  } else {
    constexpr int CH = BN / 4;             // 16-B chunks per tile row
    constexpr int NI = 256 * CH / 512;     // chunks per thread
    f32x4_t v[NI];
    if (WRITE != 1) {
#pragma unroll
      for (int t = 0; t < NI; ++t) {
        const int idx = (wave * NI + t) * 64 + lane;     // a wave instruction covers 64 consecutive chunks (row-major in the tile)
        v[t] = *(const f32x4_t*)(in + base + (size_t)(idx / CH) * ld + (idx % CH) * 4);
      }
    }
#pragma unroll
    for (int t = 0; t < NI; ++t) {
      const int idx = (wave * NI + t) * 64 + lane;
      if (WRITE == 0) acc += v[t];
      else {
        const f32x4_t o = WRITE == 1 ? (f32x4_t){1.f, 2.f, 3.f, (float)lane} : v[t];
        *(f32x4_t*)(out + base + (size_t)(idx / CH) * ld + (idx % CH) * 4) = o;
      }
    }
  }
  if (WRITE == 0 && acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}

template <int BN, int MODE, int WRITE>
static void run(const char* name, float* in, float* out, int M, int N, float* sink) {
  const int tiles_n = N / BN, grid = (M / 256) * tiles_n;
  hipEvent_t e0, e1;
  HCHECK(hipEventCreate(&e0));
  HCHECK(hipEventCreate(&e1));
  std::vector<float> us;
  for (int r = 0; r < 7; ++r) {
    HCHECK(hipEventRecord(e0, nullptr));
    hipLaunchKernelGGL((tile_io<BN, MODE, WRITE>), dim3(grid), dim3(512), 0, nullptr, in, out, N, tiles_n, sink);
    HCHECK(hipEventRecord(e1, nullptr));
    HCHECK(hipEventSynchronize(e1));
    float ms;
    HCHECK(hipEventElapsedTime(&ms, e0, e1));
    if (r >= 2) us.push_back(ms * 1e3f);
  }
  std::sort(us.begin(), us.end());
  const double bytes = (double)M * N * 4 * (WRITE == 2 ? 2 : 1);
  printf("%-34s M%6d N%5d: %8.1f us  %6.2f TB/s  (%d tiles, %.1f rounds of 256)\n", name, M, N, us[us.size() / 2], bytes / us[us.size() / 2] * 1e-6,
         grid, grid / 256.0);
}

int main() {
  const int M = 131072, N = 1280;   // 671 MB: larger than the 256-MB Infinity Cache
  float *in, *out, *sink;
  HCHECK(hipMalloc(&in, (size_t)M * N * 4));
  HCHECK(hipMalloc(&out, (size_t)M * N * 4));
  HCHECK(hipMalloc(&sink, 16));
  HCHECK(hipMemset(in, 0, (size_t)M * N * 4));
  for (int rep = 0; rep < 2; ++rep) {
    run<320, 0, 0>("read  256x320 fragment pattern", in, out, M, N, sink);
    run<320, 1, 0>("read  256x320 row pattern", in, out, M, N, sink);
    run<256, 0, 0>("read  256x256 fragment pattern", in, out, M, N, sink);
    run<256, 1, 0>("read  256x256 row pattern", in, out, M, N, sink);
    run<320, 0, 1>("write 256x320 fragment pattern", in, out, M, N, sink);
    run<320, 1, 1>("write 256x320 row pattern", in, out, M, N, sink);
    run<320, 0, 2>("copy  256x320 fragment pattern", in, out, M, N, sink);
    run<320, 1, 2>("copy  256x320 row pattern", in, out, M, N, sink);
    // one round only (what a 2-round GEMM launch sees per round)
    run<320, 0, 0>("read  frag, 1 round", in, out, 16384, N, sink);
    run<320, 1, 0>("read  row,  1 round", in, out, 16384, N, sink);
    run<320, 0, 1>("write frag, 1 round", in, out, 16384, N, sink);
    run<320, 1, 1>("write row,  1 round", in, out, 16384, N, sink);
  }
  return 0;
}
