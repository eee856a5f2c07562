// Elementwise / layout kernels (HBM-bound; vectorised 8–16 B per lane, grid-stride).
#include <stdarg.h>
#include "sx_common.h"

// ---- error plumbing (shared by all translation units) -----------------------------------------------
static thread_local char g_err[512] = "";
void sx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* sx_last_error(void) { return g_err; }
extern "C" int sx_version(void) { return 1; }

namespace sxk_elementwise {

inline dim3 grid_for(int64_t n, int per_thread = 1) {
  int64_t blocks = (n + (int64_t)256 * per_thread - 1) / ((int64_t)256 * per_thread);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  return dim3((unsigned)blocks);
}

__device__ __forceinline__ float ld_as_f32(const void* p, int dt, int64_t i) {
  if (dt == SX_F32) return ((const float*)p)[i];
  const unsigned short u = ((const unsigned short*)p)[i];
  return dt == SX_BF16 ? BF16::to_f32(u) : F16::to_f32(u);
}
__device__ __forceinline__ void st_from_f32(void* p, int dt, int64_t i, float v) {
  if (dt == SX_F32) ((float*)p)[i] = v;
  else ((unsigned short*)p)[i] = dt == SX_BF16 ? BF16::from_f32(v) : F16::from_f32(v);
}

__global__ void cast_kernel(const void* src, int sdt, void* dst, int ddt, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
#pragma unroll
    for (int e = 0; e < 4; ++e) st_from_f32(dst, ddt, 4 * i + e, ld_as_f32(src, sdt, 4 * i + e));
  }
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    st_from_f32(dst, ddt, i, ld_as_f32(src, sdt, i));
}

// x = hi + lo with hi = bf16(x), lo = bf16(x - hi): 16 mantissa bits in two bf16 MFMA operands. The VAE's fp32-grade mode
// computes A·W as Ah·Wh + Ah·Wl + Al·Wh in ONE GEMM with a tripled K: A rows are laid out [hi | hi | lo] (role 0), W rows
// [hi | lo | hi] (role 1). x - hi is exact in fp32, so the only rounding left is lo's own 2^-9 on a 2^-9-sized term.
__global__ void split_bf16_kernel(const float* x, unsigned short* out, int64_t rows, int n4, int role) {
  const int64_t total = rows * n4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / n4;
    const int q = (int)(i - r * n4);
    const f32x4_t v = ((const f32x4_t*)x)[i];
    unsigned short h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      h[e] = BF16::from_f32(v[e]);
      l[e] = BF16::from_f32(v[e] - BF16::to_f32(h[e]));
    }
    u32x2_t oh, ol;
    oh[0] = h[0] | ((unsigned)h[1] << 16); oh[1] = h[2] | ((unsigned)h[3] << 16);
    ol[0] = l[0] | ((unsigned)l[1] << 16); ol[1] = l[2] | ((unsigned)l[3] << 16);
    u32x2_t* o = (u32x2_t*)out + r * 3 * n4 + q;
    o[0] = oh;
    o[n4] = role ? ol : oh;
    o[2 * n4] = role ? oh : ol;
  }
}

__global__ void copy2d_kernel(const float* src, int64_t sld, float* dst, int64_t dld, int64_t rows, int cols4) {
  const int64_t total = rows * cols4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / cols4;
    const int c = (int)(i - r * cols4);
    *(f32x4_t*)(dst + r * dld + 4 * c) = *(const f32x4_t*)(src + r * sld + 4 * c);
  }
}

__global__ void add_kernel(const float* a, const float* b, float* y, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = a[i] + b[i];
}

// img[B][3][S][S] fp32 -> patches[B*G*G][Kpad], k = c*P*P + py*P + px (conv1 weight [width][3][P][P] flattened)
template <typename TT>
__global__ void patchify_kernel(const float* img, unsigned short* out, int B, int S, int P, int Kpad) {
  const int G = S / P, K = 3 * P * P;
  const int64_t total = (int64_t)B * G * G * Kpad;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int k = (int)(i % Kpad);
    const int64_t row = i / Kpad;
    float v = 0.f;
    if (k < K) {
      const int gx = (int)(row % G), gy = (int)((row / G) % G), b = (int)(row / ((int64_t)G * G));
      const int c = k / (P * P), rem = k % (P * P), py = rem / P, px = rem % P;
      v = img[(((int64_t)b * 3 + c) * S + (gy * P + py)) * S + gx * P + px];
    }
    out[i] = TT::from_f32(v);
  }
}

// x[B][H][W][Cin] fp32 -> out[B*H*W][Kpad], k = (ky*3 + kx)*Cin + c, zero padding outside the image
template <typename TT>
__global__ void im2col3x3_kernel(const float* x, unsigned short* out, int B, int H, int W, int Cin, int Kpad) {
  const int K = 9 * Cin;
  const int64_t total = (int64_t)B * H * W * Kpad;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int k = (int)(i % Kpad);
    const int64_t row = i / Kpad;
    float v = 0.f;
    if (k < K) {
      const int ox = (int)(row % W), oy = (int)((row / W) % H), b = (int)(row / ((int64_t)W * H));
      const int tap = k / Cin, c = k % Cin, iy = oy + tap / 3 - 1, ix = ox + tap % 3 - 1;
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[(((int64_t)b * H + iy) * W + ix) * Cin + c];
    }
    out[i] = TT::from_f32(v);
  }
}

__global__ void avgpool_tokens_kernel(const float* x, float* y, int B, int L, int D, int k) {
  const int Lo = L / k;
  const int64_t total = (int64_t)B * Lo * D;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const float inv = 1.0f / (float)k;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int d = (int)(i % D);
    const int64_t t = i / D;
    const int lo = (int)(t % Lo), b = (int)(t / Lo);
    float s = 0.f;
    for (int j = 0; j < k; ++j) s += x[((int64_t)b * L + lo * k + j) * D + d];
    y[i] = s * inv;
  }
}

__global__ void timestep_embedding_kernel(const float* t, const int* idx_dev, void* out, int n, int dim, int dt) {
  const int half = dim / 2;
  const int fixed = idx_dev ? *idx_dev : -1;
  const int total = n * half;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int j = i % half, r = i / half;
    const float f = expf(-9.210340371976184f * (float)j / (float)half);  // ln(10000)
    const float a = (fixed >= 0 ? t[fixed] : t[r]) * f;
    st_from_f32(out, dt, (int64_t)r * dim + j, cosf(a));          // flip_sin_to_cos: cos first
    st_from_f32(out, dt, (int64_t)r * dim + half + j, sinf(a));
  }
}

__global__ void nchw_to_nhwc_kernel(const float* src, float* dst, int ld, int B, int C, int HW) {
  const int64_t total = (int64_t)B * C * HW;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % C);
    const int64_t t = i / C;
    const int hw = (int)(t % HW), b = (int)(t / HW);
    dst[((int64_t)b * HW + hw) * ld + c] = src[((int64_t)b * C + c) * HW + hw];
  }
}
__global__ void nhwc_to_nchw_kernel(const float* src, int ld, float* dst, int B, int C, int HW) {
  const int64_t total = (int64_t)B * C * HW;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int hw = (int)(i % HW);
    const int64_t t = i / HW;
    const int c = (int)(t % C), b = (int)(t / C);
    dst[i] = src[((int64_t)b * HW + hw) * ld + c];
  }
}

__global__ void cfg_euler_kernel(const float* eps, float* lat, float* scaled_next, const float* sigmas,
                                 const int* step_dev, int nb, int64_t n, int C, int ld, float gs, float igs,
                                 int mode) {
  const int step = *step_dev;
  const float sigma = sigmas[step], sigma_next = sigmas[step + 1];
  const float inv_next = rsqrtf(sigma_next * sigma_next + 1.0f);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float x = lat[i];
    float e;
    if (mode == 0) {
      const float eu = eps[i], et = eps[n + i];
      e = eu + gs * (et - eu);
    } else {
      const float x0t = x - sigma * eps[i], x0i = x - sigma * eps[n + i], x0u = x - sigma * eps[2 * n + i];
      const float x0 = x0u + gs * (x0t - x0i) + igs * (x0i - x0u);
      e = (x0 - x) / (-sigma);
    }
    const float xn = x + e * (sigma_next - sigma);
    lat[i] = xn;
    if (scaled_next) {
      const float sv = xn * inv_next;
      const int64_t hw = i / C;
      const int c = (int)(i - hw * C);
      const int64_t HW = n / C;
      for (int k = 0; k < nb; ++k) scaled_next[((int64_t)k * HW + hw) * ld + c] = sv;
    }
  }
}

__global__ void add_i32_kernel(int* p, int delta, int n) {
  if ((int)threadIdx.x < n) p[threadIdx.x] += delta;
}

}  // namespace sxk_elementwise
using namespace sxk_elementwise;

#define ST ((hipStream_t)stream)

extern "C" int sx_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream) {
  SX_CHECK(src && dst && n >= 0, "sx_cast: bad args");
  if (n == 0) return SX_OK;
  hipLaunchKernelGGL(cast_kernel, grid_for(n, 4), dim3(256), 0, ST, src, src_dtype, dst, dst_dtype, n);
  SX_HIP_LAUNCH_CHECK();
  return SX_OK;
}
extern "C" int sx_split_bf16(const float* x, void* out, int64_t rows, int cols, int role, void* stream) {
  SX_CHECK(x && out && rows >= 0 && cols > 0 && cols % 4 == 0, "sx_split_bf16: bad args (cols %% 4 must be 0)");
  SX_CHECK(role == 0 || role == 1, "sx_split_bf16: role must be 0 (A rows: hi|hi|lo) or 1 (W rows: hi|lo|hi)");
  if (rows == 0) return SX_OK;
  hipLaunchKernelGGL(split_bf16_kernel, grid_for(rows * (cols / 4)), dim3(256), 0, ST, x, (unsigned short*)out, rows, cols / 4, role);
  SX_HIP_LAUNCH_CHECK();
  return SX_OK;
}
extern "C" int sx_copy2d_f32(const float* src, int64_t src_ld, float* dst, int64_t dst_ld, int64_t rows, int cols,
                             void* stream) {
  SX_CHECK(src && dst && cols % 4 == 0 && src_ld % 4 == 0 && dst_ld % 4 == 0, "sx_copy2d_f32: need multiples of 4");
  hipLaunchKernelGGL(copy2d_kernel, grid_for(rows * (cols / 4)), dim3(256), 0, ST, src, src_ld, dst, dst_ld, rows,
                     cols / 4);
  SX_HIP_LAUNCH_CHECK();
  return SX_OK;
}
extern "C" int sx_add_f32(const float* a, const float* b, float* y, int64_t n, void* stream) {
  SX_CHECK(a && b && y, "sx_add_f32: null");
  hipLaunchKernelGGL(add_kernel, grid_for(n), dim3(256), 0, ST, a, b, y, n);
  SX_HIP_LAUNCH_CHECK();
  return SX_OK;
}
extern "C" int sx_patchify(const float* img, void* patches, int B, int S, int P, int Kpad, int dtype, void* stream) {
  SX_CHECK(img && patches && S % P == 0 && Kpad >= 3 * P * P, "sx_patchify: bad geometry");
  const int64_t n = (int64_t)B * (S / P) * (S / P) * Kpad;
  if (dtype == SX_BF16)
    hipLaunchKernelGGL(patchify_kernel<BF16>, grid_for(n), dim3(256), 0, ST, img, (unsigned short*)patches, B, S, P, Kpad);
  else
    hipLaunchKernelGGL(patchify_kernel<F16>, grid_for(n), dim3(256), 0, ST, img, (unsigned short*)patches, B, S, P, Kpad);
  SX_HIP_LAUNCH_CHECK();
  return SX_OK;
}
extern "C" int sx_im2col3x3_small(const float* x, void* out, int B, int H, int W, int Cin, int Kpad, int dtype,
                                  void* stream) {
  SX_CHECK(x && out && Kpad >= 9 * Cin, "sx_im2col3x3_small: Kpad too small");
  const int64_t n = (int64_t)B * H * W * Kpad;
  if (dtype == SX_BF16)
    hipLaunchKernelGGL(im2col3x3_kernel<BF16>, grid_for(n), dim3(256), 0, ST, x, (unsigned short*)out, B, H, W, Cin, Kpad);
  else
    hipLaunchKernelGGL(im2col3x3_kernel<F16>, grid_for(n), dim3(256), 0, ST, x, (unsigned short*)out, B, H, W, Cin, Kpad);
  SX_HIP_LAUNCH_CHECK();
  return SX_OK;
}
extern "C" int sx_avgpool_tokens(const float* x, float* y, int B, int L, int D, int k, void* stream) {
  SX_CHECK(x && y && k > 0 && L % k == 0, "sx_avgpool_tokens: L %% k != 0");
  hipLaunchKernelGGL(avgpool_tokens_kernel, grid_for((int64_t)B * (L / k) * D), dim3(256), 0, ST, x, y, B, L, D, k);
  SX_HIP_LAUNCH_CHECK();
  return SX_OK;
}
extern "C" int sx_timestep_embedding(const float* t, const int32_t* idx_dev, void* out, int n, int dim, int dtype,
                                     void* stream) {
  SX_CHECK(t && out && dim % 2 == 0, "sx_timestep_embedding: dim must be even");
  hipLaunchKernelGGL(timestep_embedding_kernel, grid_for((int64_t)n * dim / 2), dim3(256), 0, ST, t, idx_dev, out, n, dim, dtype);
  SX_HIP_LAUNCH_CHECK();
  return SX_OK;
}
extern "C" int sx_nchw_to_nhwc(const float* src, float* dst, int ld, int B, int C, int HW, void* stream) {
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid_for((int64_t)B * C * HW), dim3(256), 0, ST, src, dst, ld, B, C, HW);
  SX_HIP_LAUNCH_CHECK();
  return SX_OK;
}
extern "C" int sx_nhwc_to_nchw(const float* src, int ld, float* dst, int B, int C, int HW, void* stream) {
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid_for((int64_t)B * C * HW), dim3(256), 0, ST, src, ld, dst, B, C, HW);
  SX_HIP_LAUNCH_CHECK();
  return SX_OK;
}
extern "C" int sx_cfg_euler_step(const float* eps, float* latents, float* scaled_next, const float* sigmas_dev,
                                 const int32_t* step_dev, int nb, int64_t n, int C, int ld_scaled, float gs,
                                 float igs, int mode, void* stream) {
  SX_CHECK(eps && latents && sigmas_dev && step_dev, "sx_cfg_euler_step: null pointer");
  SX_CHECK((mode == 0 && nb == 2) || (mode == 1 && nb == 3), "sx_cfg_euler_step: mode/nb mismatch");
  SX_CHECK(C > 0 && n % C == 0 && ld_scaled >= C, "sx_cfg_euler_step: C/ld");
  hipLaunchKernelGGL(cfg_euler_kernel, grid_for(n), dim3(256), 0, ST, eps, latents, scaled_next, sigmas_dev, step_dev,
                     nb, n, C, ld_scaled, gs, igs, mode);
  SX_HIP_LAUNCH_CHECK();
  return SX_OK;
}
extern "C" int sx_add_i32_n(int32_t* p, int delta, int n, void* stream) {
  SX_CHECK(p && n >= 1 && n <= 64, "sx_add_i32: bad args");
  hipLaunchKernelGGL(add_i32_kernel, dim3(1), dim3(64), 0, ST, p, delta, n);
  SX_HIP_LAUNCH_CHECK();
  return SX_OK;
}
extern "C" int sx_add_i32(int32_t* p, int delta, void* stream) { return sx_add_i32_n(p, delta, 1, stream); }

// profiling hook: an empty launch with a name of its own — tools/kstats_step.py cuts a rocprofv3 kernel trace at these dispatches so
// that a per-kernel table covers the timed step only (not the model build in front of it)
__global__ void profile_marker_kernel(int tag) { (void)tag; }
extern "C" int sx_profile_marker(int tag, void* stream) {
  hipLaunchKernelGGL(profile_marker_kernel, dim3(1), dim3(64), 0, ST, tag);
  SX_HIP_LAUNCH_CHECK();
  return SX_OK;
}

// y = silu(x) cast to 16 bit (time-embedding path: every ResnetBlock2D consumes silu(emb))
__global__ void silu_cast_kernel(const float* x, void* y, int dt, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float v = silu_f(x[i]);
    ((unsigned short*)y)[i] = dt == SX_BF16 ? BF16::from_f32(v) : F16::from_f32(v);
  }
}
extern "C" int sx_silu_cast(const float* x, void* y, int dtype, int64_t n, void* stream) {
  SX_CHECK(x && y && (dtype == SX_F16 || dtype == SX_BF16), "sx_silu_cast: bad args");
  hipLaunchKernelGGL(silu_cast_kernel, dim3((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)), dim3(256), 0,
                     ST, x, y, dtype, n);
  SX_HIP_LAUNCH_CHECK();
  return SX_OK;
}


// ---- row-sharded conv input: slab + neighbour rows + zero borders in one pass (seqpar.with_halo) ---------------------------
namespace sxk_elem_halo {
__global__ __launch_bounds__(256) void halo_pack_kernel(const u32x4_t* x, const u32x4_t* prev, const u32x4_t* next, u32x4_t* out, int B,
                                                        int Hl, int W, int c16, int off, int rows) {
  // one 16-B piece (8 channels) of the output per thread
  const long long total = (long long)B * rows * (W + off) * c16;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % c16);
    long long t = i / c16;
    const int col = (int)(t % (W + off));
    t /= (W + off);
    const int row = (int)(t % rows), b = (int)(t / rows);
    u32x4_t v = {0u, 0u, 0u, 0u};
    const int sc = col - off;
    if (sc >= 0) {
      if (row == 0) { if (prev) v = prev[((long long)b * W + sc) * c16 + c]; }
      else if (row <= Hl) v = x[(((long long)b * Hl + (row - 1)) * W + sc) * c16 + c];
      else if (next) v = next[((long long)b * W + sc) * c16 + c];
    }
    out[i] = v;
  }
}
}  // namespace sxk_elem_halo

extern "C" int sx_halo_pack(const void* x, const void* prev_row, const void* next_row, void* out, int B, int Hl, int W, int C,
                            int left_col, int bottom, void* stream) {
  SX_CHECK(x && out && B > 0 && Hl > 0 && W > 0 && C > 0 && C % 8 == 0, "sx_halo_pack: bad arguments (C %% 8 == 0)");
  const int rows = Hl + 1 + (bottom ? 1 : 0), off = left_col ? 1 : 0;
  const long long total = (long long)B * rows * (W + off) * (C / 8);
  const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
  hipLaunchKernelGGL(sxk_elem_halo::halo_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const u32x4_t*)x,
                     (const u32x4_t*)prev_row, (const u32x4_t*)(bottom ? next_row : nullptr), (u32x4_t*)out, B, Hl, W, C / 8, off, rows);
  SX_HIP_LAUNCH_CHECK();
  return SX_OK;
}
