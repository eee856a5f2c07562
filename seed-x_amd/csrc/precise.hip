// fp32-grade activations for the Llama decoder (LlamaForCausalLM(precise=True)): north_star asks for logits within 1e-3 of the
// reference; at 40 layers one 16-bit rounding per MFMA operand adds up to 2.2e-3 (tools/llm_error_budget.py: RMSNorm output 1.0e-3,
// q / k before and after RoPE 1.2e-3, v, attention output, GLU output 0.3-0.6e-3 each). The weights of a 16-bit checkpoint are exact,
// so only the ACTIVATION side needs more mantissa:
//   * every GEMM A operand travels as two 16-bit planes x = hi + lo (hi = rn16(x), lo = rn16(x - hi)): sx_gemm a_planes = 2 walks
//     K twice over the same weight tiles, sx_gemv x_planes = 2 feeds every weight fragment to two MFMAs — no extra weight bytes;
//   * q, k, v never become 16-bit: the qkv GEMM stores fp32, RoPE runs in fp32 into an fp32 KV cache (rope_kv_f32_kernel), and
//     attention (attn_f32_kernel) is fp32 FMA work — T <= a few hundred keys per head on this path, 1 % of the LLM's FLOPs.
// This file holds the kernels that only exist for that mode. It is compiled WITHOUT -ffast-math (csrc/build.sh): x - float(rn16(x))
// and the softmax bookkeeping must not be re-associated.
// Reference being matched: modeling_llama_xformer.py:95 (RMSNorm), :141-149 (RoPE), :204-239 (attention), run by the fp32 oracle.
#include "sx_common.h"
#include <type_traits>

namespace sxk_precise {

// hi saturates at the largest finite fp16 (a value beyond 65504 would round to inf and make lo = -inf): the remainder then travels in lo, so
// the pair stays exact-ish up to 2 x 65504 — real Llama checkpoints have "massive activation" channels in the thousands, not beyond. bf16 has
// the fp32 exponent range: the clamp never binds there.
template <typename TT>
__device__ __forceinline__ void split1(float x, unsigned short& hi, unsigned short& lo) {
  hi = TT::from_f32(std::is_same<TT, F16>::value ? __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f) : x);
  lo = TT::from_f32(x - TT::to_f32(hi));
}

// 8 consecutive columns c8 .. c8+7 of activation row `row` → the two planes.
//   tiled == 0: out[row][2*cols] = [hi(cols) | lo(cols)]                                   (A operand of sx_gemm, a_planes = 2)
//   tiled == nrb >= 1: operand tiles [2 planes][nrb row blocks][cols/32][16][32], hi plane first, row < 16 nrb (x operand of sx_gemv,
//                      x_planes = 2; nrb = 1: <= 16 rows, nrb = 2 (round 6): 17..32 lock-step sequences)
template <typename TT>
__device__ __forceinline__ void store_planes8(unsigned short* out, int tiled, int row, int cols, int c8, const float* v) {
  u32x4_t h, l;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    unsigned short h0, l0, h1, l1;
    split1<TT>(v[2 * e], h0, l0);
    split1<TT>(v[2 * e + 1], h1, l1);
    h[e] = (unsigned)h0 | ((unsigned)h1 << 16);
    l[e] = (unsigned)l0 | ((unsigned)l1 << 16);
  }
  if (tiled) {
    unsigned short* b = out + ((size_t)(row >> 4) * (size_t)(cols >> 5) + (size_t)(c8 >> 5)) * 512 + (size_t)(row & 15) * 32 + (c8 & 31);
    *(u32x4_t*)b = h;
    *(u32x4_t*)(b + (size_t)cols * 16 * (size_t)tiled) = l;
  } else {
    unsigned short* b = out + (size_t)row * 2 * cols + c8;
    *(u32x4_t*)b = h;
    *(u32x4_t*)(b + cols) = l;
  }
}

template <typename TT>
__global__ __launch_bounds__(256) void split16_kernel(const float* x, long long ldx, unsigned short* out, int rows, int cols, int tiled) {
  const int cpr = cols >> 3;
  const long long total = (long long)rows * cpr;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int row = (int)(i / cpr), c8 = (int)(i - (long long)row * cpr) * 8;
    const float* src = x + (size_t)row * ldx + c8;
    const f32x4_t a = *(const f32x4_t*)src, b = *(const f32x4_t*)(src + 4);
    const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    store_planes8<TT>(out, tiled, row, cols, c8, v);
  }
}

// LlamaRMSNorm (modeling_llama_xformer.py:95 → transformers 4.30.2: w * (x * rsqrt(mean(x^2) + eps)), fp32). One workgroup per row.
// Outputs: y32 (fp32, optional: the final norm's hidden states) and / or the two operand planes.
template <typename TT>
__global__ __launch_bounds__(256) void rmsnorm_planes_kernel(const float* x, const float* gamma, float* y32, unsigned short* out, int cols,
                                                             float eps, int tiled) {
  __shared__ float red[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* xr = x + (size_t)row * cols;
  float ss = 0.f;
  for (int c8 = tid * 8; c8 < cols; c8 += 2048) {
    const f32x4_t a = *(const f32x4_t*)(xr + c8), b = *(const f32x4_t*)(xr + c8 + 4);
    ss += (a[0] * a[0] + a[1] * a[1]) + (a[2] * a[2] + a[3] * a[3]) + (b[0] * b[0] + b[1] * b[1]) + (b[2] * b[2] + b[3] * b[3]);
  }
  ss = wave_sum(ss);
  if ((tid & 63) == 0) red[tid >> 6] = ss;
  __syncthreads();
  const float tot = (red[0] + red[1]) + (red[2] + red[3]);
  const float rstd = 1.0f / sqrtf(tot / (float)cols + eps);
  for (int c8 = tid * 8; c8 < cols; c8 += 2048) {
    const f32x4_t a = *(const f32x4_t*)(xr + c8), b = *(const f32x4_t*)(xr + c8 + 4);
    const f32x4_t g0 = *(const f32x4_t*)(gamma + c8), g1 = *(const f32x4_t*)(gamma + c8 + 4);
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = g0[e] * (a[e] * rstd);
      v[4 + e] = g1[e] * (b[e] * rstd);
    }
    if (y32) {
      *(f32x4_t*)(y32 + (size_t)row * cols + c8) = (f32x4_t){v[0], v[1], v[2], v[3]};
      *(f32x4_t*)(y32 + (size_t)row * cols + c8 + 4) = (f32x4_t){v[4], v[5], v[6], v[7]};
    }
    if (out) store_planes8<TT>(out, tiled, row, cols, c8, v);
  }
}

// RoPE (modeling_llama_xformer.py:141-149) of the q and k heads of fp32 qkv rows [G*T][3*H*D] (q rotated in place) + append of the
// rotated k and of v to the fp32 caches [G][H][Tmax][D] at position pos0[g] + t. The tables are rounded to the model dtype first
// (:128-131 casts them to x.dtype), the products stay fp32.
// V16 (round 6, the "mixed" cache): v is appended in the model's 16-bit dtype (vc = 16-bit [G][H][Tmax][D], same ELEMENT strides) —
// k stays fp32: the score noise of a rounded k goes through the softmax (tools/llm_error_budget.py: q / k 1.2e-3 of the 2.2e-3 at 40
// layers, v 0.6e-3), while a rounded v only perturbs the weighted average. Three quarters of the fp32 cache's bytes.
template <typename TT, bool V16>
__global__ void rope_kv_f32_kernel(float* qkv, float* kc, void* vc_, const float* cos_t, const float* sin_t, const int* pos0_dev, int T,
                                   int H, int D, int Tmax, int G, long long seq_stride) {
  const int half = D / 2;
  const int64_t total = (int64_t)G * T * H * half;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % half);
    const int h = (int)((i / half) % H);
    const int r = (int)(i / ((int64_t)half * H));
    const int g = r / T, t = r - g * T;
    int pos = pos0_dev[g] + t;
    const bool pos_ok = pos >= 0 && pos < Tmax;
    if (!pos_ok) pos = 0;
    const float c = TT::to_f32(TT::from_f32(cos_t[(size_t)pos * half + j]));
    const float s = TT::to_f32(TT::from_f32(sin_t[(size_t)pos * half + j]));
    float* row = qkv + (size_t)r * 3 * H * D;
    float* qh = row + (size_t)h * D;
    const float* kh = row + (size_t)(H + h) * D;
    const float* vh = row + (size_t)(2 * H + h) * D;
    const float q1 = qh[j], q2 = qh[j + half];
    qh[j] = q1 * c - q2 * s;
    qh[j + half] = q2 * c + q1 * s;
    if (!pos_ok) continue;                  // a device-resident position past the cache never writes outside it
    const float k1 = kh[j], k2 = kh[j + half];
    float* kd = kc + (size_t)g * seq_stride + ((size_t)h * Tmax + pos) * D;
    kd[j] = k1 * c - k2 * s;
    kd[j + half] = k2 * c + k1 * s;
    const size_t vo = (size_t)g * seq_stride + ((size_t)h * Tmax + pos) * D;
    if constexpr (V16) {
      unsigned short* vd = (unsigned short*)vc_ + vo;
      vd[j] = TT::from_f32(vh[j]);
      vd[j + half] = TT::from_f32(vh[j + half]);
    } else {
      float* vd = (float*)vc_ + vo;
      vd[j] = vh[j];
      vd[j + half] = vh[j + half];
    }
  }
}

// Causal attention over the fp32 cache, fp32 FMA arithmetic (modeling_llama_xformer.py:204-239: prefill is causal, a q_len == 1 step
// sees the whole cache — both are "row t of the chunk sees keys 0 .. pos0 + t"). Workgroup = (QB query rows of the chunk, head,
// sequence), 16 groups of 16 lanes: a group owns keys grp, grp + 16, ...; a lane owns 8 of the D <= 128 head dims; every key row is
// loaded once for the QB query rows. Online softmax per (group, row), the 16 groups meet in LDS in group order (deterministic).
// Output: the two 16-bit planes of the context rows (the o-projection's A operand), row-major or tiled (store_planes8's layouts,
// one element at a time).
struct AttnF32P {
  const float* q;          // row r = g*T + t at q + r*q_stride, head h at + h*D (already rotated)
  const float* kc;
  const void* vc;          // fp32, or (V16 kernels) the model's 16-bit dtype with the same element strides
  unsigned short* out;
  const int* pos0_dev;     // [G] cache position of the chunk's first token (causal only)
  long long q_stride, seq_stride, row_stride, head_stride;   // K / V element strides: sequence, key row, head
  int T, H, D, Tmax, tiled, causal;
  float scale;
  float* part;             // SPLIT kernels: partial results [G][H][nsplit][D + 2] = (unnormalised sum over the split's keys, running max, sum)
  int nsplit;
  // T = 1 with RoPE fused in (round 6): q is the UNROTATED row of the qkv buffer, knew / vnew the new token's k / v rows (same row stride
  // as q), cos_t / sin_t the [Tmax][D/2] fp32 tables. The workgroup (the split that owns key pos0[g]) rotates q and k, appends k / v to
  // the cache — what rope_kv_f32_kernel does in a launch of its own — and attends over keys 0 .. pos0 - 1 from the cache plus the new
  // key from registers. nullptr = q is rotated and the cache already holds the token.
  const float* cos_t;
  const float* sin_t;
  const float* knew;
  const float* vnew;
};

// QB query rows per workgroup: 4 for the decode step and short chunks, 8 for prefill (every key row is loaded once per QB rows: the
// K / V re-reads from L2 halve — 1.5k-token prompts, BASELINE config 5). DC = 8-dim chunks per lane: 1 covers D <= 128 (16 lanes x 8),
// 2 covers D <= 256 (lane dl owns dims [8 dl, 8 dl + 8) and [128 + 8 dl, 128 + 8 dl + 8): the LLM-side resamplers' head_dim 160).
// SPLIT (round 6; T = 1, QB = 1): blockIdx.x is a KEY split, not a query block — a decode step of few sequences (BASELINE config 5:
// 4 sequences x 40 heads = 160 workgroups on 256 CUs, each walking 1.5k keys in 94 dependent iterations) is latency-bound, not
// byte-bound; with the keys of a head spread over nsplit workgroups the walk is nsplit times shorter and the chip is full. Every split
// writes (sum of p v, running max, sum of p) to `part`, attn_f32_combine_kernel merges them in split order (deterministic).
template <typename TT, int QB, int DC, bool V16 = false, bool SPLIT = false>
__global__ __launch_bounds__(256) void attn_f32_kernel(const AttnF32P p) {
  constexpr int DW = 8 * DC, DMAX = 128 * DC;
  static_assert(!SPLIT || (QB == 1 && DC == 1), "key splits: the single-row decode form");
  __shared__ float red[4][QB][DMAX + 4];
  const int q0 = SPLIT ? 0 : blockIdx.x * QB, h = blockIdx.y, g = blockIdx.z, D = p.D;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = wave * 4 + (lane >> 4), dl = lane & 15;
  bool dvalid[DC];
#pragma unroll
  for (int c = 0; c < DC; ++c) dvalid[c] = c * 128 + dl * 8 < D;
  int pos0 = p.causal ? p.pos0_dev[g] : p.Tmax;   // not causal: every row sees all Tmax keys
  if (pos0 < 0) pos0 = 0;
  const int nq = min(QB, p.T - q0);
  float qv[QB][DW], o[QB][DW], m_run[QB], l_run[QB];
#pragma unroll
  for (int i = 0; i < QB; ++i) {
    m_run[i] = -INFINITY;
    l_run[i] = 0.f;
#pragma unroll
    for (int e = 0; e < DW; ++e) { qv[i][e] = 0.f; o[i][e] = 0.f; }
#pragma unroll
    for (int c = 0; c < DC; ++c) {
      if (i < nq && dvalid[c]) {
        const float* qr = p.q + (size_t)((size_t)g * p.T + q0 + i) * p.q_stride + (size_t)h * D + c * 128 + dl * 8;
        const f32x4_t a = *(const f32x4_t*)qr, b = *(const f32x4_t*)(qr + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { qv[i][8 * c + e] = a[e] * p.scale; qv[i][8 * c + 4 + e] = b[e] * p.scale; }
      }
    }
  }
  const float* kh = p.kc + (size_t)g * p.seq_stride + (size_t)h * p.head_stride;
  const size_t vbase = (size_t)g * p.seq_stride + (size_t)h * p.head_stride;
  int kend = min(p.Tmax, pos0 + q0 + nq);       // keys 0 .. kend-1 are visible to the block's last row
  int kbeg = 0;
  int chunk = kend;
  if constexpr (SPLIT) {                              // this split's share of the keys: chunks of a multiple of 16 keys
    chunk = ((kend + p.nsplit - 1) / p.nsplit + 15) & ~15;
    kbeg = (int)blockIdx.x * chunk;
    kend = min(kend, kbeg + chunk);
  }
  // ---- RoPE fused into the T = 1 launch (D = 128: lane dl holds dims 8 dl .. 8 dl + 7, its rotation partner lane dl ^ 8 the other half) ----
  bool own_new = false;                               // this workgroup attends to the NEW key (from registers) and appends it
  float knew[8], vnew[8];
  if constexpr (QB == 1 && DC == 1) {
    if (p.cos_t != nullptr) {
      const int praw = p.pos0_dev[g];
      const bool pos_ok = praw >= 0 && praw < p.Tmax;
      const int pt = pos_ok ? praw : 0, half = D >> 1;
      const int jb = (dl & 7) * 8;
      float cs[8], sn[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {                   // tables rounded to the model dtype like the reference's cast (rope_kv_f32_kernel)
        cs[e] = TT::to_f32(TT::from_f32(p.cos_t[(size_t)pt * half + jb + e]));
        sn[e] = TT::to_f32(TT::from_f32(p.sin_t[(size_t)pt * half + jb + e]));
      }
      const size_t rbase = (size_t)g * p.q_stride + (size_t)h * D;
      const float* qo = p.q + rbase + dl * 8;
      const float* qp = p.q + rbase + (dl ^ 8) * 8;
      const float* ko = p.knew + rbase + dl * 8;
      const float* kp = p.knew + rbase + (dl ^ 8) * 8;
      const float* vo = p.vnew + rbase + dl * 8;
      const float sg = (dl < 8) ? -1.0f : 1.0f;       // first half: x1 c - x2 s; second half: x2 c + x1 s
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        qv[0][e] = (qo[e] * cs[e] + sg * (qp[e] * sn[e])) * p.scale;
        knew[e] = ko[e] * cs[e] + sg * (kp[e] * sn[e]);
        vnew[e] = V16 ? TT::to_f32(TT::from_f32(vo[e])) : vo[e];
      }
      own_new = pos_ok && (SPLIT ? (praw / chunk == (int)blockIdx.x) : true);
      if (own_new && grp == 0) {                      // append: one 16-lane group covers the 128 dims
        float* kd = const_cast<float*>(kh) + (size_t)praw * p.row_stride + dl * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) kd[e] = knew[e];
        const size_t vo_c = vbase + (size_t)praw * p.row_stride + dl * 8;
        if constexpr (V16) {
          unsigned short* vd = (unsigned short*)const_cast<void*>(p.vc) + vo_c;
#pragma unroll
          for (int e = 0; e < 8; ++e) vd[e] = TT::from_f32(vo[e]);
        } else {
          float* vd = (float*)const_cast<void*>(p.vc) + vo_c;
#pragma unroll
          for (int e = 0; e < 8; ++e) vd[e] = vo[e];
        }
      }
      if (pos_ok) kend = min(kend, praw);             // the cache rows below pos; the new key comes from registers
    }
  }
  for (int t = kbeg + grp; t < kend; t += 16) {
    float kf[DW], vf[DW];
#pragma unroll
    for (int e = 0; e < DW; ++e) { kf[e] = 0.f; vf[e] = 0.f; }
#pragma unroll
    for (int c = 0; c < DC; ++c) {
      if (dvalid[c]) {
        const float* kr = kh + (size_t)t * p.row_stride + c * 128 + dl * 8;
        const size_t vo = vbase + (size_t)t * p.row_stride + c * 128 + dl * 8;
        const f32x4_t k0 = *(const f32x4_t*)kr, k1 = *(const f32x4_t*)(kr + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { kf[8 * c + e] = k0[e]; kf[8 * c + 4 + e] = k1[e]; }
        if constexpr (V16) {
          const u32x4_t w = *(const u32x4_t*)((const unsigned short*)p.vc + vo);        // 8 values in one 16-byte load
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            vf[8 * c + 2 * e] = TT::to_f32((unsigned short)(w[e] & 0xffffu));
            vf[8 * c + 2 * e + 1] = TT::to_f32((unsigned short)(w[e] >> 16));
          }
        } else {
          const float* vr = (const float*)p.vc + vo;
          const f32x4_t v0 = *(const f32x4_t*)vr, v1 = *(const f32x4_t*)(vr + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { vf[8 * c + e] = v0[e]; vf[8 * c + 4 + e] = v1[e]; }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < QB; ++i) {
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < DW; ++e) s = fmaf(kf[e], qv[i][e], s);
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      s += __shfl_xor(s, 4, 64);
      s += __shfl_xor(s, 8, 64);
      if (i < nq && t <= pos0 + q0 + i) {               // uniform over the 16-lane group
        const float m_new = fmaxf(m_run[i], s);
        const float alpha = __expf(m_run[i] - m_new);   // exp(-inf) = 0 on the first key
        const float pr = __expf(s - m_new);
        l_run[i] = l_run[i] * alpha + pr;
#pragma unroll
        for (int e = 0; e < DW; ++e) o[i][e] = fmaf(pr, vf[e], o[i][e] * alpha);
        m_run[i] = m_new;
      }
    }
  }
  if constexpr (QB == 1 && DC == 1) {
    if (own_new && grp == 0) {                        // the new token's own key / value, straight from the registers
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) s = fmaf(knew[e], qv[0][e], s);
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      s += __shfl_xor(s, 4, 64);
      s += __shfl_xor(s, 8, 64);
      const float m_new = fmaxf(m_run[0], s);
      const float alpha = __expf(m_run[0] - m_new);
      const float pr = __expf(s - m_new);
      l_run[0] = l_run[0] * alpha + pr;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[0][e] = fmaf(pr, vnew[e], o[0][e] * alpha);
      m_run[0] = m_new;
    }
  }
  // the four 16-lane groups of a wave first (xor-16 / xor-32 butterflies: the same order in every lane), then the four waves through LDS
  // in wave order: deterministic
#pragma unroll
  for (int i = 0; i < QB; ++i) {
    float mw = fmaxf(m_run[i], __shfl_xor(m_run[i], 16, 64));
    mw = fmaxf(mw, __shfl_xor(mw, 32, 64));
    const float sc = (m_run[i] == -INFINITY) ? 0.f : __expf(m_run[i] - mw);
    float lw = l_run[i] * sc;
    lw += __shfl_xor(lw, 16, 64);
    lw += __shfl_xor(lw, 32, 64);
#pragma unroll
    for (int e = 0; e < DW; ++e) {
      float t = o[i][e] * sc;
      t += __shfl_xor(t, 16, 64);
      t += __shfl_xor(t, 32, 64);
      o[i][e] = t;
    }
    if ((lane >> 4) == 0) {
#pragma unroll
      for (int c = 0; c < DC; ++c) {
        if (dvalid[c]) {
#pragma unroll
          for (int e = 0; e < 8; ++e) red[wave][i][c * 128 + dl * 8 + e] = o[i][8 * c + e];
        }
      }
      if (dl == 0) {
        red[wave][i][DMAX] = mw;
        red[wave][i][DMAX + 1] = lw;
      }
    }
  }
  __syncthreads();
  // thread → (row i, dim d): 256 threads cover QB * DMAX outputs in QB * DMAX / 256 passes
  for (int idx = threadIdx.x; idx < QB * DMAX; idx += 256) {
    const int i = idx / DMAX, d = idx - i * DMAX;
    if (i >= nq || d >= D) continue;
    float mg = -INFINITY;
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) mg = fmaxf(mg, red[gg][i][DMAX]);
    float acc = 0.f, lsum = 0.f;
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) {
      const float mgk = red[gg][i][DMAX];
      const float w = (mgk == -INFINITY) ? 0.f : __expf(mgk - mg);
      acc = fmaf(w, red[gg][i][d], acc);
      lsum = fmaf(w, red[gg][i][DMAX + 1], lsum);
    }
    if constexpr (SPLIT) {
      float* pr = p.part + ((size_t)((size_t)g * p.H + h) * p.nsplit + blockIdx.x) * (D + 2);
      pr[d] = acc;
      if (d == 0) { pr[D] = mg; pr[D + 1] = lsum; }
      continue;
    }
    const float val = lsum > 0.f ? acc / lsum : 0.f;
    unsigned short hi, lo;
    split1<TT>(val, hi, lo);
    const int cols = p.H * D, col = h * D + d;
    if (p.tiled) {
      const int row = g * p.T + q0 + i;
      unsigned short* b = p.out + ((size_t)(row >> 4) * (size_t)(cols >> 5) + (size_t)(col >> 5)) * 512 + (size_t)(row & 15) * 32 + (col & 31);
      b[0] = hi;
      b[(size_t)cols * 16 * (size_t)p.tiled] = lo;
    } else {
      unsigned short* b = p.out + (size_t)((size_t)g * p.T + q0 + i) * 2 * cols + col;
      b[0] = hi;
      b[cols] = lo;
    }
  }
}

// merge of the key splits of one (head, sequence): thread d adds the partials in split order
template <typename TT>
__global__ __launch_bounds__(128) void attn_f32_combine_kernel(const AttnF32P p) {
  const int h = blockIdx.x, g = blockIdx.y, d = threadIdx.x, D = p.D;
  if (d >= D) return;
  const float* pr = p.part + (size_t)((size_t)g * p.H + h) * p.nsplit * (D + 2);
  float mg = -INFINITY;
  for (int s = 0; s < p.nsplit; ++s) mg = fmaxf(mg, pr[(size_t)s * (D + 2) + D]);
  float acc = 0.f, lsum = 0.f;
  for (int s = 0; s < p.nsplit; ++s) {
    const float ms = pr[(size_t)s * (D + 2) + D];
    const float w = (ms == -INFINITY) ? 0.f : __expf(ms - mg);
    acc = fmaf(w, pr[(size_t)s * (D + 2) + d], acc);
    lsum = fmaf(w, pr[(size_t)s * (D + 2) + D + 1], lsum);
  }
  const float val = lsum > 0.f ? acc / lsum : 0.f;
  unsigned short hi, lo;
  split1<TT>(val, hi, lo);
  const int cols = p.H * D, col = h * D + d;
  if (p.tiled) {
    unsigned short* b = p.out + ((size_t)(g >> 4) * (size_t)(cols >> 5) + (size_t)(col >> 5)) * 512 + (size_t)(g & 15) * 32 + (col & 31);
    b[0] = hi;
    b[(size_t)cols * 16 * (size_t)p.tiled] = lo;
  } else {
    unsigned short* b = p.out + (size_t)g * 2 * cols + col;
    b[0] = hi;
    b[cols] = lo;
  }
}

// ---- the same attention on the matrix pipe, for chunks above 8 tokens (prefill, the forced image chunk) ------------------------------
// v_mfma_f32_32x32x2_f32 multiplies fp32 operands exactly and accumulates in fp32 — the arithmetic of the VALU kernel above at the fp32
// MFMA rate (64 FLOP / clk / SIMD = the whole fp32 vector rate in one instruction stream that leaves the VALU free for the softmax).
// The VALU kernel is O(T^2) FMA work on the vector pipe: 495 us per launch at T = 165 (84 launches per headline step), quadratic from
// there (BASELINE config 5 prefills 1.5k tokens).
//   wave = 32 query rows x all visible keys, flash-style over 32-key tiles; workgroup = 4 waves = 128 query rows of one (head, sequence).
//   S^T[key][q] = K · Q^T: the contraction index is split by LANE HALF, d = 64 (lane >> 5) + step — so a lane's 64 operand values are 256
//     CONTIGUOUS bytes of its own row, for Q (held in registers for the whole pass, pre-scaled) and for the K tile (16-byte loads straight
//     from the cache, no LDS: a wave's tile is its own);
//   softmax per query column: a lane holds 16 of the column's 32 scores, its partner lane ^ 32 the others (one exchange for the max, one
//     for the sum);
//   O^T[d][q] += V^T · P^T: step t contracts the t-th key of each lane half (keys 8 (t >> 2) + 4 half + (t & 3): exactly the S^T rows
//     the half holds, so P feeds the B operand without moving), the A operand is V[key][4 (lane & 31) + block .. ]: one 16-byte (fp32) or
//     8-byte (16-bit v, the mixed cache) load per step; output row index r of block b is head dim 4 r + b, so a lane ends with 16 groups of
//     4 CONSECUTIVE head dims of its query row → 8-byte plane stores.
// D = 128 only (the decoder's head_dim); everything else keeps the VALU kernel.
typedef float f32x16v_t __attribute__((ext_vector_type(16)));
template <typename TT, bool V16>
__global__ __launch_bounds__(256) void attn_f32_mfma_kernel(const AttnF32P p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int D = 128;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = blockIdx.y, g = blockIdx.z;
  const int half = lane >> 5, l32 = lane & 31;
  const int q0 = blockIdx.x * 128 + wave * 32;
  if (q0 >= p.T) return;                                       // (no barriers in this kernel)
  int pos0 = p.pos0_dev[g];
  if (pos0 < 0) pos0 = 0;
  const int qi = q0 + l32;                                      // this lane's query row (column of S^T / O^T)
  const bool qok = qi < p.T;
  const int qc = qok ? qi : p.T - 1;
  // Q operand: 64 contiguous floats of the lane's row, pre-scaled
  float qv[64];
  {
    const float* qr = p.q + (size_t)((size_t)g * p.T + qc) * p.q_stride + (size_t)h * D + half * 64;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const f32x4_t a = *(const f32x4_t*)(qr + 4 * e);
      qv[4 * e] = a[0] * p.scale; qv[4 * e + 1] = a[1] * p.scale; qv[4 * e + 2] = a[2] * p.scale; qv[4 * e + 3] = a[3] * p.scale;
    }
  }
  const float* kh = p.kc + (size_t)g * p.seq_stride + (size_t)h * p.head_stride;
  const size_t vbase = (size_t)g * p.seq_stride + (size_t)h * p.head_stride;
  f32x16v_t o[4];
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int j = 0; j < 16; ++j) o[b][j] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const int q_last = min(p.T, q0 + 32) - 1;
  const int kend = min(p.Tmax, pos0 + q_last + 1);              // keys 0 .. kend - 1 are visible to the wave's last row
  const int vis = pos0 + qi;                                    // last key this lane's row may see
  for (int k0 = 0; k0 < kend; k0 += 32) {
    // ---- S^T tile = K[k0 .. k0+31] · Q^T ----
    const int kr_ = min(k0 + l32, p.Tmax - 1);
    const float* kr = kh + (size_t)kr_ * p.row_stride + half * 64;
    f32x16v_t sacc;
#pragma unroll
    for (int j = 0; j < 16; ++j) sacc[j] = 0.f;
    f32x4_t kq[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) kq[e] = *(const f32x4_t*)(kr + 4 * e);
#pragma unroll
    for (int e = 0; e < 16; ++e)
#pragma unroll
      for (int c = 0; c < 4; ++c) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kq[e][c], qv[4 * e + c], sacc, 0, 0, 0);
    // ---- online softmax of the lane's column (16 local keys + the partner's 16) ----
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int key = k0 + (j >> 2) * 8 + half * 4 + (j & 3);
      sacc[j] = (key <= vis && key < kend) ? sacc[j] : -INFINITY;
      mx = fmaxf(mx, sacc[j]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = (m_new == -INFINITY) ? 1.f : __expf(m_run - m_new);      // exp(-inf) = 0 on the first visible tile
    float ls = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float pr = (sacc[j] == -INFINITY) ? 0.f : __expf(sacc[j] - m_new);
      sacc[j] = pr;
      ls += pr;
    }
    ls += __shfl_xor(ls, 32, 64);
    l_run = l_run * alpha + ls;
    m_run = m_new;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int j = 0; j < 16; ++j) o[b][j] *= alpha;
    // ---- O^T += V^T · P^T ----
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int key = min(k0 + (t >> 2) * 8 + half * 4 + (t & 3), p.Tmax - 1);
      const size_t vo = vbase + (size_t)key * p.row_stride + 4 * l32;
      float vv[4];
      if constexpr (V16) {
        const u32x2_t w = *(const u32x2_t*)((const unsigned short*)p.vc + vo);
        vv[0] = TT::to_f32((unsigned short)(w[0] & 0xffffu)); vv[1] = TT::to_f32((unsigned short)(w[0] >> 16));
        vv[2] = TT::to_f32((unsigned short)(w[1] & 0xffffu)); vv[3] = TT::to_f32((unsigned short)(w[1] >> 16));
      } else {
        const f32x4_t w = *(const f32x4_t*)((const float*)p.vc + vo);
        vv[0] = w[0]; vv[1] = w[1]; vv[2] = w[2]; vv[3] = w[3];
      }
#pragma unroll
      for (int b = 0; b < 4; ++b) o[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[b], sacc[t], o[b], 0, 0, 0);
    }
  }
  // ---- normalise, split into planes, store: lane = query row qi, register j of block b = head dim 4 r_j + b ----
  if (!qok) return;
  const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
  const int cols = p.H * D;
  unsigned short* orow = p.out + (size_t)((size_t)g * p.T + qi) * 2 * cols + (size_t)h * D;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int r = (j >> 2) * 8 + half * 4 + (j & 3);
    unsigned short hi[4], lo[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) split1<TT>(o[b][j] * inv, hi[b], lo[b]);
    u32x2_t wh, wl;
    wh[0] = (unsigned)hi[0] | ((unsigned)hi[1] << 16); wh[1] = (unsigned)hi[2] | ((unsigned)hi[3] << 16);
    wl[0] = (unsigned)lo[0] | ((unsigned)lo[1] << 16); wl[1] = (unsigned)lo[2] | ((unsigned)lo[3] << 16);
    *(u32x2_t*)(orow + 4 * r) = wh;
    *(u32x2_t*)(orow + cols + 4 * r) = wl;
  }
#endif
}

static dim3 gs_grid(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > 65535 * 4) b = 65535 * 4;
  if (b < 1) b = 1;
  return dim3((unsigned)b);
}

}  // namespace sxk_precise
using namespace sxk_precise;

#define ST ((hipStream_t)stream)

extern "C" int sx_split16(const float* x, int64_t ldx, void* out, int rows, int cols, int dtype, void* stream) {
  SX_CHECK(x && out, "sx_split16: null pointer");
  const int tiled = (dtype & SX_TILED16) ? (rows + 15) / 16 : 0, dt = dtype & 0xff;     // = the tiled layout's row blocks
  SX_CHECK(dt == SX_F16 || dt == SX_BF16, "sx_split16: dtype");
  SX_CHECK(rows >= 1 && cols >= 8 && cols % 8 == 0 && ldx >= cols && ldx % 4 == 0 && (((uintptr_t)x) & 15) == 0 && (((uintptr_t)out) & 15) == 0,
           "sx_split16: rows=%d cols=%d ldx=%lld (cols %% 8, 16-B aligned rows)", rows, cols, (long long)ldx);
  SX_CHECK(!tiled || (rows <= 32 && cols % 32 == 0), "sx_split16: operand tiles hold <= 32 rows of cols %% 32 == 0");
  const int64_t n = (int64_t)rows * (cols / 8);
  if (dt == SX_BF16) hipLaunchKernelGGL(split16_kernel<BF16>, gs_grid(n), dim3(256), 0, ST, x, (long long)ldx, (unsigned short*)out, rows, cols, tiled);
  else hipLaunchKernelGGL(split16_kernel<F16>, gs_grid(n), dim3(256), 0, ST, x, (long long)ldx, (unsigned short*)out, rows, cols, tiled);
  SX_HIP_LAUNCH_CHECK();
  return SX_OK;
}

extern "C" int sx_rmsnorm_planes(const float* x, const float* gamma, float* y32, void* out16, int rows, int cols, float eps, int dtype,
                                 void* stream) {
  SX_CHECK(x && gamma && (y32 || out16), "sx_rmsnorm_planes: null pointer");
  const int tiled = (dtype & SX_TILED16) ? (rows + 15) / 16 : 0, dt = dtype & 0xff;
  SX_CHECK(dt == SX_F16 || dt == SX_BF16, "sx_rmsnorm_planes: dtype");
  SX_CHECK(rows >= 1 && cols >= 8 && cols % 8 == 0 && (((uintptr_t)x) & 15) == 0 && (((uintptr_t)gamma) & 15) == 0,
           "sx_rmsnorm_planes: rows=%d cols=%d (cols %% 8, 16-B aligned)", rows, cols);
  SX_CHECK(!tiled || !out16 || (rows <= 32 && cols % 32 == 0), "sx_rmsnorm_planes: operand tiles hold <= 32 rows of cols %% 32 == 0");
  if (dt == SX_BF16)
    hipLaunchKernelGGL(rmsnorm_planes_kernel<BF16>, dim3(rows), dim3(256), 0, ST, x, gamma, y32, (unsigned short*)out16, cols, eps, tiled);
  else
    hipLaunchKernelGGL(rmsnorm_planes_kernel<F16>, dim3(rows), dim3(256), 0, ST, x, gamma, y32, (unsigned short*)out16, cols, eps, tiled);
  SX_HIP_LAUNCH_CHECK();
  return SX_OK;
}

static int rope_kv_f32_impl(float* qkv, float* kcache, void* vcache, int v16, const float* cos_tab, const float* sin_tab,
                            const int32_t* pos0_dev, int G, int T, int H, int D, int Tmax, int64_t cache_seq_stride, int table_dtype,
                            void* stream) {
  SX_CHECK(qkv && kcache && vcache && cos_tab && sin_tab && pos0_dev, "sx_rope_kv_append_f32: null pointer");
  SX_CHECK(D % 2 == 0 && G >= 1 && T >= 1 && H >= 1, "sx_rope_kv_append_f32: D/G/T/H");
  SX_CHECK(table_dtype == SX_F16 || table_dtype == SX_BF16, "sx_rope_kv_append_f32: table_dtype");
  const int64_t n = (int64_t)G * T * H * (D / 2);
#define SX_ROPE_GO(TT, V16)                                                                                                            \
  hipLaunchKernelGGL((rope_kv_f32_kernel<TT, V16>), gs_grid(n), dim3(256), 0, ST, qkv, kcache, vcache, cos_tab, sin_tab, pos0_dev, T, H, D, \
                     Tmax, G, (long long)cache_seq_stride)
  if (table_dtype == SX_BF16) { if (v16) SX_ROPE_GO(BF16, true); else SX_ROPE_GO(BF16, false); }
  else { if (v16) SX_ROPE_GO(F16, true); else SX_ROPE_GO(F16, false); }
#undef SX_ROPE_GO
  SX_HIP_LAUNCH_CHECK();
  return SX_OK;
}

extern "C" int sx_rope_kv_append_f32(float* qkv, float* kcache, float* vcache, const float* cos_tab, const float* sin_tab,
                                     const int32_t* pos0_dev, int G, int T, int H, int D, int Tmax, int64_t cache_seq_stride,
                                     int table_dtype, void* stream) {
  return rope_kv_f32_impl(qkv, kcache, vcache, 0, cos_tab, sin_tab, pos0_dev, G, T, H, D, Tmax, cache_seq_stride, table_dtype, stream);
}

extern "C" int sx_rope_kv_append_f32_v16(float* qkv, float* kcache, void* vcache16, const float* cos_tab, const float* sin_tab,
                                         const int32_t* pos0_dev, int G, int T, int H, int D, int Tmax, int64_t cache_seq_stride,
                                         int table_dtype, void* stream) {
  return rope_kv_f32_impl(qkv, kcache, vcache16, 1, cos_tab, sin_tab, pos0_dev, G, T, H, D, Tmax, cache_seq_stride, table_dtype, stream);
}

static int g_attn_f32_mfma = 1;    // 0: chunks above 8 tokens keep the VALU kernel (A/B and the bit-reference of the tests)
extern "C" int sx_attention_f32_variant(int v) {
  SX_CHECK(v == 0 || v == 1, "sx_attention_f32_variant: 0 (VALU kernels only) or 1 (fp32 MFMA kernel for chunks above 8 tokens)");
  g_attn_f32_mfma = v;
  return SX_OK;
}

extern "C" int sx_attention_f32(const sx_attn_f32_args* a, void* stream) {
  SX_CHECK(a && a->q && a->kcache && a->vcache && a->out && (a->pos0_dev || !a->causal), "sx_attention_f32: null pointer");
  const int tiled = (a->dtype & SX_TILED16) ? (int)(((int64_t)a->G * a->T + 15) / 16) : 0, dt = a->dtype & 0xff;   // row blocks of the tiled output
  SX_CHECK(dt == SX_F16 || dt == SX_BF16, "sx_attention_f32: dtype (of the output planes)");
  SX_CHECK(a->D % 8 == 0 && a->D >= 8 && a->D <= 256, "sx_attention_f32: head_dim %d (multiple of 8, <= 256)", a->D);
  SX_CHECK(a->G >= 1 && a->T >= 1 && a->H >= 1 && a->Tmax >= 1, "sx_attention_f32: G/T/H/Tmax");
  SX_CHECK(a->q_row_stride >= (int64_t)a->H * a->D && a->q_row_stride % 4 == 0 && (((uintptr_t)a->q) & 15) == 0, "sx_attention_f32: q_row_stride");
  SX_CHECK(!tiled || ((int64_t)a->G * a->T <= 32 && (a->H * a->D) % 32 == 0), "sx_attention_f32: operand tiles hold <= 32 rows, H*D %% 32 == 0");
  AttnF32P p;
  p.q = a->q; p.kc = a->kcache; p.out = (unsigned short*)a->out; p.pos0_dev = a->pos0_dev;
  p.q_stride = a->q_row_stride; p.seq_stride = a->cache_seq_stride;
  p.row_stride = a->kv_row_stride > 0 ? a->kv_row_stride : a->D;
  p.head_stride = a->kv_head_stride > 0 ? a->kv_head_stride : (int64_t)a->Tmax * a->D;
  p.vc = a->vcache;
  SX_CHECK(p.row_stride % 4 == 0 && p.head_stride % 4 == 0 && p.seq_stride % 4 == 0 && (((uintptr_t)a->kcache) & 15) == 0 &&
           (((uintptr_t)a->vcache) & 15) == 0, "sx_attention_f32: K / V strides and pointers must keep 16-B alignment");
  p.T = a->T; p.H = a->H; p.D = a->D; p.Tmax = a->Tmax; p.tiled = tiled; p.scale = a->scale; p.causal = a->causal ? 1 : 0;
  p.part = a->scratch; p.nsplit = a->nsplit;
  p.cos_t = a->rope_cos; p.sin_t = a->rope_sin; p.knew = a->k_new; p.vnew = a->v_new;
  if (a->rope_cos) {
    SX_CHECK(a->T == 1 && a->causal && a->D == 128 && a->rope_sin && a->k_new && a->v_new && a->kv_row_stride <= 0 + (int64_t)a->D,
             "sx_attention_f32: the fused RoPE form is the T = 1 causal step at head_dim 128 over the cache layout (rope_sin, k_new, v_new set)");
    SX_CHECK((((uintptr_t)a->k_new) & 15) == 0 && (((uintptr_t)a->v_new) & 15) == 0, "sx_attention_f32: k_new / v_new alignment");
  }
  if (a->T == 1 && a->nsplit > 1) {
    // the decode step of few sequences: key splits + combine (two launches)
    SX_CHECK(a->scratch && a->causal && a->D <= 128 && a->nsplit <= 64 && (!a->v16 || (p.row_stride % 8 == 0 && p.head_stride % 8 == 0 && p.seq_stride % 8 == 0)),
             "sx_attention_f32: key splits need scratch [G][H][nsplit][D + 2] floats, a causal T = 1 call and head_dim <= 128");
    p.vc = a->vcache;
    const dim3 grid(a->nsplit, a->H, a->G);
#define SX_SPLIT_GO(TT, V16) hipLaunchKernelGGL((attn_f32_kernel<TT, 1, 1, V16, true>), grid, dim3(256), 0, ST, p)
    if (dt == SX_BF16) { if (a->v16) SX_SPLIT_GO(BF16, true); else SX_SPLIT_GO(BF16, false); }
    else { if (a->v16) SX_SPLIT_GO(F16, true); else SX_SPLIT_GO(F16, false); }
#undef SX_SPLIT_GO
    SX_HIP_LAUNCH_CHECK();
    if (dt == SX_BF16) hipLaunchKernelGGL(attn_f32_combine_kernel<BF16>, dim3(a->H, a->G), dim3(128), 0, ST, p);
    else hipLaunchKernelGGL(attn_f32_combine_kernel<F16>, dim3(a->H, a->G), dim3(128), 0, ST, p);
    SX_HIP_LAUNCH_CHECK();
    return SX_OK;
  }
  const bool mfma = g_attn_f32_mfma && a->causal && a->T > 8 && a->D == 128 && !tiled && (((uintptr_t)a->out) & 7) == 0 &&
                    p.row_stride % 4 == 0 && a->q_row_stride % 4 == 0;
  if (mfma) {
    p.vc = a->vcache;
    const dim3 grid((a->T + 127) / 128, a->H, a->G);
    if (a->v16) {
      SX_CHECK(p.row_stride % 8 == 0 && p.head_stride % 8 == 0 && p.seq_stride % 8 == 0, "sx_attention_f32: 16-bit V rows must be 16-B aligned");
      if (dt == SX_BF16) hipLaunchKernelGGL((attn_f32_mfma_kernel<BF16, true>), grid, dim3(256), 0, ST, p);
      else hipLaunchKernelGGL((attn_f32_mfma_kernel<F16, true>), grid, dim3(256), 0, ST, p);
    } else {
      if (dt == SX_BF16) hipLaunchKernelGGL((attn_f32_mfma_kernel<BF16, false>), grid, dim3(256), 0, ST, p);
      else hipLaunchKernelGGL((attn_f32_mfma_kernel<F16, false>), grid, dim3(256), 0, ST, p);
    }
    SX_HIP_LAUNCH_CHECK();
    return SX_OK;
  }
  if (a->v16) {
    // mixed cache: V in the planes' 16-bit dtype (head_dim <= 128: the decoder's; the resamplers' fp32 views keep the fp32 form)
    SX_CHECK(a->v16 == 1 && a->D <= 128 && p.row_stride % 8 == 0 && p.head_stride % 8 == 0 && p.seq_stride % 8 == 0,
             "sx_attention_f32: a 16-bit V cache needs head_dim <= 128 and 16-B aligned rows");
    if (a->T > 8) {
      const dim3 grid((a->T + 7) / 8, a->H, a->G);
      if (dt == SX_BF16) hipLaunchKernelGGL((attn_f32_kernel<BF16, 8, 1, true>), grid, dim3(256), 0, ST, p);
      else hipLaunchKernelGGL((attn_f32_kernel<F16, 8, 1, true>), grid, dim3(256), 0, ST, p);
    } else if (a->T == 1) {       // the decode step: ONE query row per workgroup (QB = 4 computes four rows' dot products per key for one valid row)
      const dim3 grid(1, a->H, a->G);
      if (dt == SX_BF16) hipLaunchKernelGGL((attn_f32_kernel<BF16, 1, 1, true>), grid, dim3(256), 0, ST, p);
      else hipLaunchKernelGGL((attn_f32_kernel<F16, 1, 1, true>), grid, dim3(256), 0, ST, p);
    } else {
      const dim3 grid((a->T + 3) / 4, a->H, a->G);
      if (dt == SX_BF16) hipLaunchKernelGGL((attn_f32_kernel<BF16, 4, 1, true>), grid, dim3(256), 0, ST, p);
      else hipLaunchKernelGGL((attn_f32_kernel<F16, 4, 1, true>), grid, dim3(256), 0, ST, p);
    }
    SX_HIP_LAUNCH_CHECK();
    return SX_OK;
  }
  if (a->D > 128) {             // two 8-dim chunks per lane (the LLM-side resamplers' head_dim 160): 4 rows per workgroup
    const dim3 grid((a->T + 3) / 4, a->H, a->G);
    if (dt == SX_BF16) hipLaunchKernelGGL((attn_f32_kernel<BF16, 4, 2>), grid, dim3(256), 0, ST, p);
    else hipLaunchKernelGGL((attn_f32_kernel<F16, 4, 2>), grid, dim3(256), 0, ST, p);
  } else if (a->T > 8) {
    const dim3 grid((a->T + 7) / 8, a->H, a->G);
    if (dt == SX_BF16) hipLaunchKernelGGL((attn_f32_kernel<BF16, 8, 1>), grid, dim3(256), 0, ST, p);
    else hipLaunchKernelGGL((attn_f32_kernel<F16, 8, 1>), grid, dim3(256), 0, ST, p);
  } else if (a->T == 1) {
    const dim3 grid(1, a->H, a->G);
    if (dt == SX_BF16) hipLaunchKernelGGL((attn_f32_kernel<BF16, 1, 1>), grid, dim3(256), 0, ST, p);
    else hipLaunchKernelGGL((attn_f32_kernel<F16, 1, 1>), grid, dim3(256), 0, ST, p);
  } else {
    const dim3 grid((a->T + 3) / 4, a->H, a->G);
    if (dt == SX_BF16) hipLaunchKernelGGL((attn_f32_kernel<BF16, 4, 1>), grid, dim3(256), 0, ST, p);
    else hipLaunchKernelGGL((attn_f32_kernel<F16, 4, 1>), grid, dim3(256), 0, ST, p);
  }
  SX_HIP_LAUNCH_CHECK();
  return SX_OK;
}
