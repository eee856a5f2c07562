// LayerNorm / RMSNorm / GroupNorm(+SiLU) for gfx950. HBM-bound row reductions: vector loads,
// wave-shuffle reductions, fp32 statistics (fp64 cross-block accumulation for GroupNorm).
#include "sx_common.h"

namespace sxk_norm {

template <int IN_DT>
__device__ __forceinline__ f32x4_t load4(const void* base, size_t idx4) {
  if (IN_DT == SX_F32) {
    return ((const f32x4_t*)base)[idx4];
  } else {
    const u32x2_t u = ((const u32x2_t*)base)[idx4];
    f32x4_t v;
    if (IN_DT == SX_BF16) {
      v[0] = BF16::to_f32(u[0] & 0xffff); v[1] = BF16::to_f32(u[0] >> 16);
      v[2] = BF16::to_f32(u[1] & 0xffff); v[3] = BF16::to_f32(u[1] >> 16);
    } else {
      v[0] = F16::to_f32(u[0] & 0xffff); v[1] = F16::to_f32(u[0] >> 16);
      v[2] = F16::to_f32(u[1] & 0xffff); v[3] = F16::to_f32(u[1] >> 16);
    }
    return v;
  }
}
__device__ __forceinline__ void store4(void* base, int dt, size_t idx4, f32x4_t v) {
  if (dt == SX_F32) {
    ((f32x4_t*)base)[idx4] = v;
  } else {
    u32x2_t o;
    if (dt == SX_BF16) { o[0] = pack2<BF16>(v[0], v[1]); o[1] = pack2<BF16>(v[2], v[3]); }
    else { o[0] = pack2<F16>(v[0], v[1]); o[1] = pack2<F16>(v[2], v[3]); }
    ((u32x2_t*)base)[idx4] = o;
  }
}

// One wave per row, 4 rows per 256-thread block. The whole row is loaded ONCE into registers (NV float4 per lane,
// all loads in flight together), statistics come from two wave reductions, then the row is written: one HBM read
// and one write per element, no dependent re-read passes.
template <int IN_DT, int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const void* x, void* y, int out_dt, const float* gamma,
                                                        const float* beta, int rows, int cols, float eps, int rms) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int n4 = cols >> 2;
  const size_t base4 = (size_t)row * n4;
  f32x4_t v[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int i = lane + 64 * k;
    v[k] = (i < n4) ? load4<IN_DT>(x, base4 + i) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }
  float mean = 0.f;
  if (!rms) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
    mean = wave_sum(s) / (float)cols;
  }
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int i = lane + 64 * k;
    if (i < n4) {
      const f32x4_t d = v[k] - mean;
      q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)cols + eps);
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int i = lane + 64 * k;
    if (i < n4) {
      f32x4_t o = (v[k] - mean) * rstd * ((const f32x4_t*)gamma)[i];
      if (beta) o += ((const f32x4_t*)beta)[i];
      store4(y, out_dt, base4 + i, o);
    }
  }
}

// Few rows (single-token decode: rows == 1): one 256-thread BLOCK per row, x / gamma / beta loads all in flight
// together, two LDS block reductions. Cuts the dependent-latency chain of the wave-per-row kernel (20 loads per lane).
template <int IN_DT>
__global__ __launch_bounds__(256) void layernorm_block_kernel(const void* x, void* y, int out_dt, const float* gamma,
                                                              const float* beta, int rows, int cols, float eps, int rms) {
  __shared__ float red[8];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n4 = cols >> 2;
  const size_t base4 = (size_t)row * n4;
  constexpr int NV = 6;
  f32x4_t v[NV], g[NV], b[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int i = tid + 256 * k;
    const bool ok = i < n4;
    v[k] = ok ? load4<IN_DT>(x, base4 + i) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
    g[k] = ok ? ((const f32x4_t*)gamma)[i] : (f32x4_t){0.f, 0.f, 0.f, 0.f};
    b[k] = (ok && beta) ? ((const f32x4_t*)beta)[i] : (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }
  float mean = 0.f;
  if (!rms) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
    s = wave_sum(s);
    if (lane == 0) red[w] = s;
    __syncthreads();
    mean = (red[0] + red[1] + red[2] + red[3]) / (float)cols;
  }
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    if (tid + 256 * k < n4) {
      const f32x4_t d = v[k] - mean;
      q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
    }
  }
  q = wave_sum(q);
  if (lane == 0) red[4 + w] = q;
  __syncthreads();
  const float rstd = rsqrtf((red[4] + red[5] + red[6] + red[7]) / (float)cols + eps);
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int i = tid + 256 * k;
    if (i < n4) {
      // SX_TILED16: element (row, col) lives at tile col/32, row `row`, column col%32 of [cols/32][16][32]
      // (rows 16..31 form a second block of tiles behind the first: [rows/16][cols/32][16][32])
      const size_t o4 = (out_dt & SX_TILED16) ? ((size_t)(row >> 4) * (size_t)cols * 16 + (size_t)(i >> 3) * 512 + (size_t)(row & 15) * 32 + (size_t)(i & 7) * 4) >> 2 : base4 + i;
      store4(y, out_dt & 0xff, o4, (v[k] - mean) * rstd * g[k] + b[k]);
    }
  }
}

static int g_gn_stat_blocks = 0;   // tuning hook (sx_norm_tune): block count of the GroupNorm statistics pass, 0 = by size

// ---- GroupNorm over NHWC fp32 -----------------------------------------------------------------
constexpr int GN_MAX_SLOTS2 = 6;  // C/2 pairs per row / 256 threads  (C <= 3072)
constexpr int GN_MAX_SLOTS4 = 3;  // C/4 quads per row / 256 threads

// blockDim.x = T (320 / 160 / 256, chosen so the C/2 channel pairs tile the block evenly); each thread owns fixed
// channel pairs and walks the block's rows four at a time (4 independent 8-B loads in flight per slot).
__global__ void gn_zero_kernel(double* p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0.0;
}

// TWO: the logical [HW][C] input is the channel concatenation of x[HW][C1] and x2[HW][C - C1] (UNet up blocks:
// torch.cat([hidden, skip], dim=1) — never materialised)
template <bool TWO>
__global__ __launch_bounds__(320) void gn_stats_kernel(const float* x, const float* x2, int C1, double* stats, int HW, int C,
                                                       int groups, int rows_per_block) {
  // each thread owns fixed channel QUADS (16-B loads: 8-B accesses stream at 0.54-0.70x the 16-B rate on gfx950) and walks
  // the block's rows four at a time. A quad may straddle two groups (C = 320: 10 channels per group), a PAIR cannot
  // (channels per group are even), so the two halves of the quad are accumulated separately.
  __shared__ __attribute__((aligned(16))) float part[GN_MAX_SLOTS4 * 320 * 4];  // [channel quad][4]: every thread's own slot
  const int T = blockDim.x;
  const int b = blockIdx.y;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(HW, r0 + rows_per_block);
  const int nq = C >> 2, cpg = C / groups;
  float sl[GN_MAX_SLOTS4], ql[GN_MAX_SLOTS4], sh[GN_MAX_SLOTS4], qh[GN_MAX_SLOTS4];
  const f32x4_t* src[GN_MAX_SLOTS4];
  int ld[GN_MAX_SLOTS4];
  const int nq1 = TWO ? (C1 >> 2) : nq, nq2 = nq - nq1;
  const f32x4_t* xb = (const f32x4_t*)(x + (size_t)b * HW * (TWO ? C1 : C));
  const f32x4_t* xb2 = TWO ? (const f32x4_t*)(x2 + (size_t)b * HW * (C - C1)) : nullptr;
#pragma unroll
  for (int k = 0; k < GN_MAX_SLOTS4; ++k) {
    sl[k] = ql[k] = sh[k] = qh[k] = 0.f;
    const int qd = threadIdx.x + k * T;
    const bool second = TWO && qd >= nq1;
    src[k] = second ? xb2 + (qd - nq1) : xb + qd;
    ld[k] = second ? nq2 : nq1;
  }
  auto add = [&](int k, const f32x4_t v) {
    sl[k] += v[0] + v[1];
    ql[k] += v[0] * v[0] + v[1] * v[1];
    sh[k] += v[2] + v[3];
    qh[k] += v[2] * v[2] + v[3] * v[3];
  };
  int r = r0;
  for (; r + 3 < r1; r += 4) {
#pragma unroll
    for (int k = 0; k < GN_MAX_SLOTS4; ++k) {
      if (threadIdx.x + k * T < nq) {
        const f32x4_t v0 = src[k][(size_t)r * ld[k]], v1 = src[k][(size_t)(r + 1) * ld[k]];
        const f32x4_t v2 = src[k][(size_t)(r + 2) * ld[k]], v3 = src[k][(size_t)(r + 3) * ld[k]];
        add(k, v0); add(k, v1); add(k, v2); add(k, v3);
      }
    }
  }
  for (; r < r1; ++r) {
#pragma unroll
    for (int k = 0; k < GN_MAX_SLOTS4; ++k)
      if (threadIdx.x + k * T < nq) add(k, src[k][(size_t)r * ld[k]]);
  }
  // FIXED-ORDER reduction inside the block (round 6; LDS float atomics in arrival order before: the block's partial sums differed in
  // their last bits from run to run, and with them — through the 16-bit roundings behind every normalisation — the whole forward):
  // every thread parks its quads' four sums in its own LDS slot, one thread per (group, moment) adds the group's channel PAIRS in
  // index order (a pair never straddles a group). The blocks meet in global memory by fp64 atomics: a sum of fp32-valued terms is
  // exact in fp64 (terms spanning < 2^29), hence order-independent.
#pragma unroll
  for (int k = 0; k < GN_MAX_SLOTS4; ++k) {
    const int qd = threadIdx.x + k * T;
    if (qd < nq) *(f32x4_t*)(part + 4 * qd) = (f32x4_t){sl[k], ql[k], sh[k], qh[k]};      // pair 2 qd: (sum, sq), pair 2 qd + 1: (sum, sq)
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * groups; i += T) {
    const int gi = i >> 1, which = i & 1;
    const int p0 = gi * (cpg >> 1), p1 = p0 + (cpg >> 1);
    float a = 0.f;
    for (int pr = p0; pr < p1; ++pr) a += part[2 * pr + which];
    atomicAdd(&stats[(size_t)b * groups * 2 + i], (double)a);
  }
}

// apply: y = (x - mean) * rstd * gamma + beta (+ SiLU), optional 16-bit raw copy. Per-channel scale / shift are built ONCE per
// block in LDS (fp64 statistics → fp32 pair), then the block streams its rows as one flat run of float4 (a block's rows are
// contiguous in memory), 4 loads in flight per thread — every one of the 256 threads is busy whatever C is (C = 320 / 640
// left 31-62 % of the lanes idle when threads were tied to channel quads).
constexpr int GN_MAX_C = 3072;
template <bool TWO>
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* x, const float* x2, int C1, void* y, void* raw16, int out_dt,
                                                       const float* gamma, const float* beta, const double* stats,
                                                       int HW, int C, int groups, float eps, int silu,
                                                       int rows_per_block, int hw_total) {
  __shared__ __attribute__((aligned(16))) float s_sc[GN_MAX_C], s_sh[GN_MAX_C];
  const int b = blockIdx.y;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(HW, r0 + rows_per_block);
  const int n4 = C >> 2, cpg = C / groups;
  const double cnt = (double)hw_total * cpg;                // hw_total > HW: the statistics cover rows held by other ranks too
  for (int c = threadIdx.x; c < C; c += 256) {
    const int g = c / cpg;
    const double su = stats[((size_t)b * groups + g) * 2], sq = stats[((size_t)b * groups + g) * 2 + 1];
    const double mean = su / cnt;
    double var = sq / cnt - mean * mean;
    if (var < 0) var = 0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float ga = gamma[c] * rstd;
    s_sc[c] = ga;
    s_sh[c] = beta[c] - (float)mean * ga;
  }
  __syncthreads();
  const size_t base4 = ((size_t)b * HW + r0) * n4;          // first float4 of this block's rows
  const int total = (r1 - r0) * n4;
  const f32x4_t* x4 = (const f32x4_t*)x + (TWO ? 0 : base4);
  const f32x4_t* x4b = (const f32x4_t*)x2;
  const int n4a = C1 >> 2, n4b = n4 - n4a;
  const size_t row_base = (size_t)b * HW + r0;
  // TWO: quad (row, q) of the logical concatenation lives in x (q < n4a) or x2; (row, q) are tracked incrementally
  auto fetch = [&](int i, int row, int q) -> f32x4_t {
    if (!TWO) return x4[i];
    return q < n4a ? x4[(row_base + row) * n4a + q] : x4b[(row_base + row) * n4b + (q - n4a)];
  };
  // SX_BF16X3: bf16 planes of the fp32 value, rows laid out [hi | hi | lo] (3*C columns) — the A operand of the VAE's
  // fp32-grade GEMMs (see sx_split_bf16), written here instead of an fp32 tensor plus a separate split pass
  auto store_planes = [&](void* dst, int row, int qi, const f32x4_t v) {
    unsigned short h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      h[e] = BF16::from_f32(v[e]);
      l[e] = BF16::from_f32(v[e] - BF16::to_f32(h[e]));
    }
    u32x2_t oh, ol;
    oh[0] = h[0] | ((unsigned)h[1] << 16); oh[1] = h[2] | ((unsigned)h[3] << 16);
    ol[0] = l[0] | ((unsigned)l[1] << 16); ol[1] = l[2] | ((unsigned)l[3] << 16);
    u32x2_t* o = (u32x2_t*)dst + (row_base + row) * 3 * n4 + qi;
    o[0] = oh;
    o[n4] = oh;
    o[2 * n4] = ol;
  };
  // SX_F16X2 (round 6): fp16 planes x = hi + lo, rows laid out [hi | lo] (2*C columns) — the A operand of the VAE's fp32-grade convs
  // when their weights are exact in fp16 (a VAE loaded `.to(dtype=torch.float16)`, which is what the reference up-casts): one weight
  // plane, so A·W = Ah·W + Al·W is TWO products instead of the three of the bf16 form. hi saturates at the largest finite fp16 (the
  // remainder travels in lo); only used behind a GroupNorm, whose outputs are bounded.
  auto store_planes2 = [&](void* dst, int row, int qi, const f32x4_t v) {
    unsigned short h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      h[e] = F16::from_f32(__builtin_amdgcn_fmed3f(v[e], -65504.f, 65504.f));
      l[e] = F16::from_f32(v[e] - F16::to_f32(h[e]));
    }
    u32x2_t oh, ol;
    oh[0] = h[0] | ((unsigned)h[1] << 16); oh[1] = h[2] | ((unsigned)h[3] << 16);
    ol[0] = l[0] | ((unsigned)l[1] << 16); ol[1] = l[2] | ((unsigned)l[3] << 16);
    u32x2_t* o = (u32x2_t*)dst + (row_base + row) * 2 * n4 + qi;
    o[0] = oh;
    o[n4] = ol;
  };
  auto one = [&](int i, int row, int qi, const f32x4_t v) {
    f32x4_t o = v * *(const f32x4_t*)(s_sc + 4 * qi) + *(const f32x4_t*)(s_sh + 4 * qi);
    if (silu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = silu_f(o[e]);
    }
    if (out_dt == SX_BF16X3) {
      store_planes(y, row, qi, o);
      if (raw16) store_planes(raw16, row, qi, v);
      return;
    }
    if (out_dt == SX_F16X2) {
      store_planes2(y, row, qi, o);
      if (raw16) store_planes2(raw16, row, qi, v);
      return;
    }
    store4(y, out_dt, base4 + i, o);
    if (raw16) store4(raw16, out_dt, base4 + i, v);
  };
  const int step = 256 % n4, dstep = 256 / n4;              // (row, quad) advance by (256 / n4, 256 mod n4) per stride
  int qi = threadIdx.x % n4, ri = threadIdx.x / n4;
  int i = threadIdx.x;
  auto adv = [&](int& r_, int& q_) { q_ += step; r_ += dstep; if (q_ >= n4) { q_ -= n4; ++r_; } };
  for (; i + 768 < total; i += 1024) {
    int q1 = qi, r1_ = ri; adv(r1_, q1);
    int q2 = q1, r2_ = r1_; adv(r2_, q2);
    int q3 = q2, r3_ = r2_; adv(r3_, q3);
    const f32x4_t v0 = fetch(i, ri, qi), v1 = fetch(i + 256, r1_, q1), v2 = fetch(i + 512, r2_, q2), v3 = fetch(i + 768, r3_, q3);
    one(i, ri, qi, v0);
    one(i + 256, r1_, q1, v1);
    one(i + 512, r2_, q2, v2);
    one(i + 768, r3_, q3, v3);
    qi = q3; ri = r3_; adv(ri, qi);
  }
  for (; i < total; i += 256) {
    one(i, ri, qi, fetch(i, ri, qi));
    adv(ri, qi);
  }
}

// ---- row softmax (VAE mid-block attention: one 512-wide head over 16384 pixels, scores materialised by the GEMM) ---------------
// y[r][c] = exp(scale·x[r][c] − max_r) / Σ_c …, fp32 in → 16-bit out. Block per row, three streaming passes over the row
// (64 KB at C = 16384: it stays in L2 between passes), float4 loads, block reductions through LDS.
template <typename TT, bool F32OUT = false>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* x, long ldx, unsigned short* y, long ldy, int C,
                                                           float scale) {
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float4* xr = (const float4*)(x + (size_t)blockIdx.x * ldx);
  unsigned short* yr = y + (size_t)blockIdx.x * ldy;
  const int n4 = C >> 2;
  const float sl2 = scale * 1.4426950408889634f;   // exp(s·x − m) = exp2(s·log2e·x − m')
  float m = -INFINITY;
  for (int i = tid; i < n4; i += 256) {
    const float4 v = xr[i];
    m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
  }
  m = wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * sl2;   // scale > 0
  __syncthreads();
  float sum = 0.f;
  for (int i = tid; i < n4; i += 256) {
    const float4 v = xr[i];
    sum += exp2f(v.x * sl2 - m) + exp2f(v.y * sl2 - m) + exp2f(v.z * sl2 - m) + exp2f(v.w * sl2 - m);
  }
  sum = wave_sum(sum);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  const float inv = 1.f / (red[0] + red[1] + red[2] + red[3]);
  for (int i = tid; i < n4; i += 256) {
    const float4 v = xr[i];
    if (F32OUT) {   // ldy counts floats here
      ((float4*)((float*)y + (size_t)blockIdx.x * ldy))[i] =
          make_float4(exp2f(v.x * sl2 - m) * inv, exp2f(v.y * sl2 - m) * inv, exp2f(v.z * sl2 - m) * inv, exp2f(v.w * sl2 - m) * inv);
      continue;
    }
    u32x2_t o;
    o[0] = pack2<TT>(exp2f(v.x * sl2 - m) * inv, exp2f(v.y * sl2 - m) * inv);
    o[1] = pack2<TT>(exp2f(v.z * sl2 - m) * inv, exp2f(v.w * sl2 - m) * inv);
    *(u32x2_t*)(yr + 4 * i) = o;
  }
}

}  // namespace sxk_norm
using namespace sxk_norm;

extern "C" int sx_layernorm(const void* x, int in_dtype, void* y, int out_dtype, const float* gamma,
                            const float* beta, int rows, int cols, float eps, int rms, void* stream) {
  SX_CHECK(x && y && gamma, "sx_layernorm: null pointer");
  SX_CHECK(rows > 0 && cols > 0 && cols % 4 == 0, "sx_layernorm: rows=%d cols=%d (cols %% 4 must be 0)", rows, cols);
  const bool tiled = (out_dtype & SX_TILED16) != 0;
  SX_CHECK(!tiled || (rows <= 32 && cols % 32 == 0 && ((out_dtype & 0xff) == SX_F16 || (out_dtype & 0xff) == SX_BF16)),
           "sx_layernorm: SX_TILED16 needs rows <= 32, cols %% 32 == 0 and a 16-bit output (rows=%d cols=%d)", rows, cols);
  SX_CHECK((out_dtype & 0xff) >= SX_F16 && (out_dtype & 0xff) <= SX_F32 && (out_dtype & ~0x1ff) == 0, "sx_layernorm: bad out dtype");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((rows + 3) / 4), block(256);
  const int n4 = cols / 4;
  SX_CHECK(n4 <= 64 * 24, "sx_layernorm: cols=%d exceeds the register-resident limit (6144)", cols);
#define SX_LN_GO(DT, NV) \
  hipLaunchKernelGGL((layernorm_kernel<DT, NV>), grid, block, 0, st, x, y, out_dtype, gamma, beta, rows, cols, eps, rms)
#define SX_LN_NV(DT)                        \
  if (n4 <= 64 * 3) { SX_LN_GO(DT, 3); }    \
  else if (n4 <= 64 * 5) { SX_LN_GO(DT, 5); }  \
  else if (n4 <= 64 * 8) { SX_LN_GO(DT, 8); }  \
  else if (n4 <= 64 * 16) { SX_LN_GO(DT, 16); } \
  else { SX_LN_GO(DT, 24); }
  if (rows <= 32) {  // decode-sized: block per row
    const dim3 g1(rows);
    switch (in_dtype) {
      case SX_F32: hipLaunchKernelGGL(layernorm_block_kernel<SX_F32>, g1, block, 0, st, x, y, out_dtype, gamma, beta, rows, cols, eps, rms); break;
      case SX_F16: hipLaunchKernelGGL(layernorm_block_kernel<SX_F16>, g1, block, 0, st, x, y, out_dtype, gamma, beta, rows, cols, eps, rms); break;
      case SX_BF16: hipLaunchKernelGGL(layernorm_block_kernel<SX_BF16>, g1, block, 0, st, x, y, out_dtype, gamma, beta, rows, cols, eps, rms); break;
      default: SX_FAIL("sx_layernorm: bad in dtype %d", in_dtype);
    }
    SX_HIP_LAUNCH_CHECK();
    return SX_OK;
  }
  switch (in_dtype) {
    case SX_F32: SX_LN_NV(SX_F32) break;
    case SX_F16: SX_LN_NV(SX_F16) break;
    case SX_BF16: SX_LN_NV(SX_BF16) break;
    default:
      SX_FAIL("sx_layernorm: bad in dtype %d", in_dtype);
  }
#undef SX_LN_NV
#undef SX_LN_GO
  SX_HIP_LAUNCH_CHECK();
  return SX_OK;
}

// phase 0: statistics + apply; 1: zero + statistics of the local rows only (caller all-reduces `stats` across ranks);
// 2: apply only, `stats` holding the sums over hw_total rows per sample.
static int groupnorm_impl(const float* x, const float* x2, int C1, void* y, void* raw16, int out_dtype, const float* gamma,
                          const float* beta, double* stats, int B, int HW, int hw_total, int C, int groups, float eps,
                          int silu, int phase, void* stream) {
  SX_CHECK(x && stats, "sx_groupnorm: null pointer");
  SX_CHECK(phase == 1 || (y && gamma && beta), "sx_groupnorm: null pointer");
  SX_CHECK(!x2 || (C1 > 0 && C1 < C && C1 % 4 == 0 && (C - C1) % 4 == 0), "sx_groupnorm2: C1=%d of C=%d", C1, C);
  SX_CHECK(phase == 1 || out_dtype == SX_F16 || out_dtype == SX_BF16 || out_dtype == SX_BF16X3 || out_dtype == SX_F16X2 ||
               (out_dtype == SX_F32 && !raw16),
           "sx_groupnorm: output must be 16-bit, SX_BF16X3, SX_F16X2, or fp32 without a raw copy");
  SX_CHECK(groups > 0 && groups <= 64 && C % groups == 0, "sx_groupnorm: C=%d groups=%d", C, groups);
  SX_CHECK(C % 4 == 0 && (C / groups) % 2 == 0, "sx_groupnorm: C %% 4 and (C/groups) %% 2 must be 0 (C=%d)", C);
  SX_CHECK(C / 2 <= 256 * GN_MAX_SLOTS2 && C <= GN_MAX_C, "sx_groupnorm: C=%d too large", C);
  SX_CHECK(hw_total >= HW && HW > 0, "sx_groupnorm: hw_total=%d < HW=%d", hw_total, HW);
  hipStream_t st = (hipStream_t)stream;
  // ~2048 blocks over the chip
  int rows_per_block = (HW * B + 2047) / 2048;
  if (rows_per_block < 4) rows_per_block = 4;
  const dim3 grid((HW + rows_per_block - 1) / rows_per_block, B), block(256);
  const int nq = C / 4;
  const int T = (nq % 320 == 0) ? 320 : ((nq % 160 == 0) ? 160 : ((nq <= 128) ? 128 : 256));
  SX_CHECK((nq + T - 1) / T <= GN_MAX_SLOTS4, "sx_groupnorm: C=%d too large", C);
  if (phase != 2) {
    // zero the fp64 accumulators with a KERNEL node, not hipMemsetAsync: under hipGraph replay the memset node was not
    // ordered against the neighbouring kernel nodes — the chain after it ran concurrently with its producers (15.7 ms per
    // 250-ms UNet step "faster", reading stale activations; NaN statistics whenever a replay started on an idle GPU)
    hipLaunchKernelGGL(gn_zero_kernel, dim3((2 * B * groups + 255) / 256), dim3(256), 0, st, stats, 2 * B * groups);
    SX_HIP_LAUNCH_CHECK();
    // stats: ~512 blocks in total — every block ends with 2*groups fp64 atomics on the same B*groups*2 words, so the
    // block count (not the byte count) bounds this kernel once the atomics serialise in L2
    // block count: ~256 KB of fp32 input per block, 512..2048 blocks (tools/lab/gn_lab, profiles/r3_gn_lab.log: 512 blocks were
    // right for the 32x32 / 64x64 levels, 3.4 TB/s instead of 6.0 at 128x128 x 320 and 1.7 instead of 3.6 on the VAE's 1024^2 x 128)
    long nblk = g_gn_stat_blocks;
    if (nblk <= 0) {
      nblk = ((long)B * HW * C * 4) >> 18;
      nblk = nblk < 512 ? 512 : (nblk > 2048 ? 2048 : nblk);
    }
    int rows_stats = (int)(((long)HW * B + nblk - 1) / nblk);
    if (rows_stats < 8) rows_stats = 8;
    const dim3 grid_s((HW + rows_stats - 1) / rows_stats, B);
    if (x2)
      hipLaunchKernelGGL(gn_stats_kernel<true>, grid_s, dim3(T), 0, st, x, x2, C1, stats, HW, C, groups, rows_stats);
    else
      hipLaunchKernelGGL(gn_stats_kernel<false>, grid_s, dim3(T), 0, st, x, x2, C, stats, HW, C, groups, rows_stats);
    SX_HIP_LAUNCH_CHECK();
  }
  if (phase != 1) {
    if (x2)
      hipLaunchKernelGGL(gn_apply_kernel<true>, grid, block, 0, st, x, x2, C1, y, raw16, out_dtype, gamma, beta, stats, HW, C,
                         groups, eps, silu, rows_per_block, hw_total);
    else
      hipLaunchKernelGGL(gn_apply_kernel<false>, grid, block, 0, st, x, x2, C, y, raw16, out_dtype, gamma, beta, stats, HW, C,
                         groups, eps, silu, rows_per_block, hw_total);
    SX_HIP_LAUNCH_CHECK();
  }
  return SX_OK;
}

extern "C" int sx_norm_tune(int key, int value) {   // tuning hook: key 0 = block count of the GroupNorm statistics pass
  if (key == 0 && (value == 0 || (value >= 64 && value <= 65536))) { g_gn_stat_blocks = value; return SX_OK; }
  SX_FAIL("sx_norm_tune: unknown key %d / value %d", key, value);
}

extern "C" int sx_groupnorm2(const float* x, const float* x2, int C1, void* y, void* raw16, int out_dtype, const float* gamma,
                             const float* beta, double* stats, int B, int HW, int C, int groups, float eps, int silu,
                             void* stream) {
  return groupnorm_impl(x, x2, C1, y, raw16, out_dtype, gamma, beta, stats, B, HW, HW, C, groups, eps, silu, 0, stream);
}

extern "C" int sx_groupnorm_sp(const float* x, const float* x2, int C1, void* y, void* raw16, int out_dtype, const float* gamma,
                               const float* beta, double* stats, int B, int HW, int hw_total, int C, int groups, float eps,
                               int silu, int phase, void* stream) {
  SX_CHECK(phase == 1 || phase == 2, "sx_groupnorm_sp: phase must be 1 (statistics) or 2 (apply)");
  return groupnorm_impl(x, x2, C1, y, raw16, out_dtype, gamma, beta, stats, B, HW, hw_total, C, groups, eps, silu, phase, stream);
}

extern "C" int sx_groupnorm(const float* x, void* y, void* raw16, int out_dtype, const float* gamma,
                            const float* beta, double* stats, int B, int HW, int C, int groups, float eps, int silu,
                            void* stream) {
  return sx_groupnorm2(x, nullptr, 0, y, raw16, out_dtype, gamma, beta, stats, B, HW, C, groups, eps, silu, stream);
}

extern "C" int sx_softmax_rows(const float* x, int64_t ldx, void* y, int64_t ldy, int rows, int cols, float scale,
                               int out_dtype, void* stream) {
  SX_CHECK(x && y, "sx_softmax_rows: null pointer");
  SX_CHECK(rows > 0 && cols > 0 && cols % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0, "sx_softmax_rows: rows=%d cols=%d", rows, cols);
  SX_CHECK(scale > 0.f, "sx_softmax_rows: scale must be positive");
  SX_CHECK(out_dtype >= SX_F16 && out_dtype <= SX_F32, "sx_softmax_rows: bad output dtype");
  hipStream_t st = (hipStream_t)stream;
  if (out_dtype == SX_F32)
    hipLaunchKernelGGL((softmax_rows_kernel<BF16, true>), dim3(rows), dim3(256), 0, st, x, (long)ldx, (unsigned short*)y, (long)ldy, cols, scale);
  else if (out_dtype == SX_BF16)
    hipLaunchKernelGGL(softmax_rows_kernel<BF16>, dim3(rows), dim3(256), 0, st, x, (long)ldx, (unsigned short*)y, (long)ldy, cols, scale);
  else
    hipLaunchKernelGGL(softmax_rows_kernel<F16>, dim3(rows), dim3(256), 0, st, x, (long)ldx, (unsigned short*)y, (long)ldy, cols, scale);
  SX_HIP_LAUNCH_CHECK();
  return SX_OK;
}
