// Image pre/post-processing on the GPU: the byte/integer work that sits either side of the three dense
// paths in the reference's scripts, restated as HBM-bound gfx950 kernels so that a request enters the GPU as the raw
// uint8 image and leaves it as the uint8 image.
//
//   * sx_resample_u8      PIL `Image.resize` (Pillow's ImagingResample, 8 bits per channel): separable antialiased
//                         bilinear / bicubic convolution in 22-bit fixed point, horizontal pass then vertical pass, each
//                         pass rounded and clipped to uint8 — bit-exact with Pillow. Called by the reference at
//                         src/inference/any_res.py:88-114,183,187 (`image.resize`, default BICUBIC) and, through
//                         torchvision `transforms.Resize` (BILINEAR), at src/processer/transforms.py:5-20.
//                         The coefficient tables (Pillow's precompute_coeffs + normalize_coeffs_8bpc, double precision)
//                         are built by the host and passed in; the device does only integer arithmetic.
//   * sx_u8_to_chw_lut    crop (any_res.py:117-136 `divide_to_patches`) + ToTensor + Normalize (transforms.py:16-19) as
//                         ONE gather through a 3x256 float table built by the host in float32 (so the values equal
//                         torch's `(u/255 - mean)/std` bit for bit whatever the GPU's division does)
//   * sx_chw_to_u8_image  VaeImageProcessor.postprocess [ext] as used at pipeline_stable_diffusion_xl_t2i_edit.py:986:
//                         (x/2 + 0.5).clamp(0,1) → HWC → (·255).round() → uint8
//   * sx_marker_mask      ids_cmp_mask construction of eval_img2text_seed_x_i.py:153-160: True strictly between the k-th
//                         <img>/<patch> and the k-th </img>/</patch>
//   * sx_l2norm_dim1      F.normalize(x) with its default dim=1 (the TOKEN axis) of ResamplerXLV2(normalize=True),
//                         resampler.py:271-272
#include "sx_common.h"

namespace sxk_preproc {

#define ST ((hipStream_t)stream)
constexpr int PRECISION_BITS = 32 - 8 - 2;  // Pillow Resample.c

__device__ __forceinline__ unsigned char clip8(int v) {
  v >>= PRECISION_BITS;  // arithmetic shift, like Pillow's lookup index
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// one thread per output pixel (all channels). src rows: [y][x][C] with a byte stride; output dense [rows][Wout][C].
template <int C>
__global__ void resample_h_kernel(const unsigned char* __restrict__ src, int64_t src_stride, int y_first, int rows,
                                  int Wout, const int* __restrict__ kk, const int* __restrict__ bounds, int ksize,
                                  unsigned char* __restrict__ dst) {
  const int64_t n = (int64_t)rows * Wout;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int y = (int)(i / Wout), xo = (int)(i - (int64_t)y * Wout);
    const int xmin = bounds[2 * xo], cnt = bounds[2 * xo + 1];
    const int* k = kk + (int64_t)xo * ksize;
    const unsigned char* p = src + (int64_t)(y_first + y) * src_stride + (int64_t)xmin * C;
    int acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 1 << (PRECISION_BITS - 1);
    for (int t = 0; t < cnt; ++t) {
      const int w = k[t];
#pragma unroll
      for (int c = 0; c < C; ++c) acc[c] += (int)p[t * C + c] * w;
    }
#pragma unroll
    for (int c = 0; c < C; ++c) dst[i * C + c] = clip8(acc[c]);
  }
}

template <int C>
__global__ void resample_v_kernel(const unsigned char* __restrict__ src, int64_t src_stride, int y_offset, int W,
                                  int Hout, const int* __restrict__ kk, const int* __restrict__ bounds, int ksize,
                                  unsigned char* __restrict__ dst) {
  const int64_t n = (int64_t)Hout * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int yo = (int)(i / W), x = (int)(i - (int64_t)yo * W);
    const int ymin = bounds[2 * yo] - y_offset, cnt = bounds[2 * yo + 1];
    const int* k = kk + (int64_t)yo * ksize;
    const unsigned char* p = src + (int64_t)ymin * src_stride + (int64_t)x * C;
    int acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 1 << (PRECISION_BITS - 1);
    for (int t = 0; t < cnt; ++t) {
      const int w = k[t];
#pragma unroll
      for (int c = 0; c < C; ++c) acc[c] += (int)p[(int64_t)t * src_stride + c] * w;
    }
#pragma unroll
    for (int c = 0; c < C; ++c) dst[i * C + c] = clip8(acc[c]);
  }
}

__global__ void u8_to_chw_lut_kernel(const unsigned char* __restrict__ src, int64_t src_stride, int x0, int y0, int Hc,
                                     int Wc, const float* __restrict__ lut, float* __restrict__ dst) {
  __shared__ float s_lut[3 * 256];
  for (int i = threadIdx.x; i < 3 * 256; i += blockDim.x) s_lut[i] = lut[i];
  __syncthreads();
  const int64_t n = (int64_t)Hc * Wc;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int y = (int)(i / Wc), x = (int)(i - (int64_t)y * Wc);
    const unsigned char* p = src + (int64_t)(y0 + y) * src_stride + (int64_t)(x0 + x) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) dst[(int64_t)c * n + i] = s_lut[c * 256 + p[c]];
  }
}

__global__ void chw_to_u8_image_kernel(const float* __restrict__ src, int64_t HW, unsigned char* __restrict__ dst) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float v = src[(int64_t)c * HW + i] / 2.0f + 0.5f;
      v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);      // NaN propagates through clamp in torch; the VAE never emits one
      dst[i * 3 + c] = (unsigned char)__builtin_rintf(v * 255.0f);
    }
  }
}

// mask[i] = min(#opens strictly before i, npairs) > #closes at or before i, npairs = min(#opens, #closes): exactly the
// reference's `for boi, eoi in zip(boi_indices, eoi_indices): mask[boi+1:eoi] = True` for sorted index lists.
__global__ void marker_mask_kernel(const int64_t* __restrict__ ids, int T, int64_t boi, int64_t bop, int64_t eoi,
                                   int64_t eop, unsigned char* __restrict__ mask) {
  extern __shared__ int s[];  // [2][blockDim]: per-thread chunk counts of opens / closes
  const int nt = blockDim.x, tid = threadIdx.x;
  const int chunk = (T + nt - 1) / nt, b = tid * chunk, e = (b + chunk < T) ? b + chunk : T;
  int no = 0, nc = 0;
  for (int i = b; i < e; ++i) {
    no += (ids[i] == boi || ids[i] == bop);
    nc += (ids[i] == eoi || ids[i] == eop);
  }
  s[tid] = no;
  s[nt + tid] = nc;
  __syncthreads();
  int po = 0, pc = 0, to = 0, tc = 0;
  for (int j = 0; j < nt; ++j) {  // T <= a few thousand: a serial scan of <= 1024 counters per thread is negligible
    if (j < tid) { po += s[j]; pc += s[nt + j]; }
    to += s[j];
    tc += s[nt + j];
  }
  const int npairs = to < tc ? to : tc;
  for (int i = b; i < e; ++i) {
    pc += (ids[i] == eoi || ids[i] == eop);                    // closes at or before i
    const int a = po < npairs ? po : npairs;                   // opens strictly before i, capped by the zip()
    mask[i] = a > pc;
    po += (ids[i] == boi || ids[i] == bop);
  }
}

// y[b][t][d] = x[b][t][d] / max(||x[b][:][d]||_2, eps): one thread per (b, d) column, coalesced over d
__global__ void l2norm_dim1_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int T, int D, float eps) {
  const int64_t n = (int64_t)B * D;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / D), d = (int)(i - (int64_t)b * D);
    const float* p = x + (int64_t)b * T * D + d;
    float ss = 0.f;
    for (int t = 0; t < T; ++t) ss += p[(int64_t)t * D] * p[(int64_t)t * D];
    float nrm = __builtin_sqrtf(ss);
    nrm = nrm > eps ? nrm : eps;
    float* q = y + (int64_t)b * T * D + d;
    for (int t = 0; t < T; ++t) q[(int64_t)t * D] = p[(int64_t)t * D] / nrm;
  }
}

inline dim3 grid_for(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return dim3((unsigned)b);
}

}  // namespace sxk_preproc
using namespace sxk_preproc;

extern "C" int sx_resample_u8(const void* src, int Hin, int Win, int C, int64_t src_stride, void* dst, int Hout, int Wout,
                              const int32_t* kk_h, const int32_t* bounds_h, int ksize_h, const int32_t* kk_v,
                              const int32_t* bounds_v, int ksize_v, int y_first, int y_rows, void* tmp, void* stream) {
  SX_CHECK(src && dst, "sx_resample_u8: null pointer");
  SX_CHECK(C == 3 || C == 1, "sx_resample_u8: C=%d (1 or 3 channels)", C);
  SX_CHECK(Hin > 0 && Win > 0 && Hout > 0 && Wout > 0, "sx_resample_u8: empty image");
  SX_CHECK(src_stride >= (int64_t)Win * C, "sx_resample_u8: src_stride");
  const bool need_h = kk_h != nullptr, need_v = kk_v != nullptr;
  SX_CHECK(need_h || need_v, "sx_resample_u8: nothing to do (Pillow returns a copy for an identity resize)");
  SX_CHECK(!need_h || (bounds_h && ksize_h > 0), "sx_resample_u8: horizontal tables");
  SX_CHECK(!need_v || (bounds_v && ksize_v > 0), "sx_resample_u8: vertical tables");
  SX_CHECK(need_h || Wout == Win, "sx_resample_u8: Wout != Win without a horizontal pass");
  SX_CHECK(need_v || Hout == Hin, "sx_resample_u8: Hout != Hin without a vertical pass");
  if (need_h && need_v) {
    SX_CHECK(tmp && y_first >= 0 && y_rows > 0 && y_first + y_rows <= Hin, "sx_resample_u8: tmp rows [%d,+%d) of %d", y_first,
             y_rows, Hin);
  }
#define SX_RS(CC)                                                                                                         \
  if (need_h) {                                                                                                           \
    const int rows = need_v ? y_rows : Hin, y0 = need_v ? y_first : 0;                                                    \
    void* out = need_v ? tmp : dst;                                                                                       \
    hipLaunchKernelGGL(resample_h_kernel<CC>, grid_for((int64_t)rows * Wout), dim3(256), 0, ST, (const unsigned char*)src, \
                       src_stride, y0, rows, Wout, kk_h, bounds_h, ksize_h, (unsigned char*)out);                         \
    SX_HIP_LAUNCH_CHECK();                                                                                                \
  }                                                                                                                       \
  if (need_v) {                                                                                                           \
    const unsigned char* in = need_h ? (const unsigned char*)tmp : (const unsigned char*)src;                             \
    const int64_t stride = need_h ? (int64_t)Wout * CC : src_stride;                                                      \
    hipLaunchKernelGGL(resample_v_kernel<CC>, grid_for((int64_t)Hout * Wout), dim3(256), 0, ST, in, stride,               \
                       need_h ? y_first : 0, Wout, Hout, kk_v, bounds_v, ksize_v, (unsigned char*)dst);                   \
    SX_HIP_LAUNCH_CHECK();                                                                                                \
  }
  if (C == 3) { SX_RS(3) } else { SX_RS(1) }
#undef SX_RS
  return SX_OK;
}

extern "C" int sx_u8_to_chw_lut(const void* src, int H, int W, int64_t src_stride, int x0, int y0, int Hc, int Wc,
                                const float* lut3x256, float* dst, void* stream) {
  SX_CHECK(src && lut3x256 && dst, "sx_u8_to_chw_lut: null pointer");
  SX_CHECK(x0 >= 0 && y0 >= 0 && Hc > 0 && Wc > 0 && x0 + Wc <= W && y0 + Hc <= H, "sx_u8_to_chw_lut: crop box [%d,%d)+(%d,%d) outside %dx%d",
           x0, y0, Wc, Hc, W, H);
  SX_CHECK(src_stride >= (int64_t)W * 3, "sx_u8_to_chw_lut: src_stride");
  hipLaunchKernelGGL(u8_to_chw_lut_kernel, grid_for((int64_t)Hc * Wc), dim3(256), 0, ST, (const unsigned char*)src, src_stride,
                     x0, y0, Hc, Wc, lut3x256, dst);
  SX_HIP_LAUNCH_CHECK();
  return SX_OK;
}

extern "C" int sx_chw_to_u8_image(const float* src, int H, int W, void* dst, void* stream) {
  SX_CHECK(src && dst && H > 0 && W > 0, "sx_chw_to_u8_image: bad args");
  hipLaunchKernelGGL(chw_to_u8_image_kernel, grid_for((int64_t)H * W), dim3(256), 0, ST, src, (int64_t)H * W, (unsigned char*)dst);
  SX_HIP_LAUNCH_CHECK();
  return SX_OK;
}

extern "C" int sx_marker_mask(const int64_t* ids, int T, int64_t boi, int64_t bop, int64_t eoi, int64_t eop, void* mask,
                              void* stream) {
  SX_CHECK(ids && mask && T >= 0, "sx_marker_mask: bad args");
  if (T == 0) return SX_OK;
  const int nt = 256;
  hipLaunchKernelGGL(marker_mask_kernel, dim3(1), dim3(nt), 2 * nt * sizeof(int), ST, ids, T, boi, bop, eoi, eop,
                     (unsigned char*)mask);
  SX_HIP_LAUNCH_CHECK();
  return SX_OK;
}

extern "C" int sx_l2norm_dim1(const float* x, float* y, int B, int T, int D, float eps, void* stream) {
  SX_CHECK(x && y && B > 0 && T > 0 && D > 0, "sx_l2norm_dim1: bad args");
  hipLaunchKernelGGL(l2norm_dim1_kernel, grid_for((int64_t)B * D), dim3(256), 0, ST, x, y, B, T, D, eps);
  SX_HIP_LAUNCH_CHECK();
  return SX_OK;
}
