// Single-token decode path of the Llama-style LLM on gfx950: HBM-bound weight-streaming GEMV, split-KV
// decode attention, RoPE + KV-cache append, embedding gather/scatter and the fused greedy/logits-rule step.
// All loop state (position, current token) lives in DEVICE memory so a whole token step is hipGraph-replayable
// with zero host round trips (the reference syncs ≥45× per token).
#include "sx_common.h"
#include <type_traits>

namespace sxk_decode {

// ---- GEMV: one wave per 2 weight rows (or one GLU pair), 16-B weight loads, x re-read through L1/L2 ----------
struct GemvP {
  const unsigned short* x;
  const unsigned short* W;
  void* y;
  const float* residual;
  int M, N, K, out_dtype, act, glu;
  int packed;   // W is in the decode layout [N/16][K/32][16 rows][32 k] (one MFMA operand tile = 1 KB contiguous)
  int y_tiled;  // y is written as operand tiles [n_out/32][16][32] (16-bit outputs)
  int x_tiled;  // x is in operand tiles [K/32][16 rows][32 k] (one 1-KB tile per MFMA B operand; rows >= M are padding)
  unsigned* ws_cnt;   // split-K: arrival counters [N/16] (zero between launches)
  float* ws_part;     //          partial sums [S][16][N]
  // RMSNorm fold (include/seedx_hip.h sx_gemv_args): producer side x16_out / ssq_out, consumer side ssq_in
  unsigned short* x16_out;
  float* ssq_out;
  const float* ssq_in;
  int ssq_parts;
  float ssq_inv_dim, ssq_eps;
  int planes2;  // MB = 2 kernels with M <= 16: x block 1 is the LO plane of an fp32-grade activation (sx_gemv_args.x_planes = 2) — the two
                // blocks' results are added ahead of the epilogue
  int out_planes;          // the 16-bit tiled output (y_tiled) or x16_out is written as two planes: block 0 = hi, block 1 = lo (rows <= 16)
  const float* x16_gamma;  // x16_out holds o * gamma[col] (the NEXT RMSNorm's gamma applied on the activation side: the weights stay exact)
};

template <typename TT, int MR>
__global__ __launch_bounds__(256) void gemv_kernel(const GemvP p) {
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);  // wave id = pair index
  int r0, r1, n_out;
  if (p.glu) {
    // 32-row groups [16 linear | 16 gate]; pair j of group g = rows (32g + j, 32g + 16 + j)
    const int g = wid >> 4, j = wid & 15;
    r0 = g * 32 + j;
    r1 = r0 + 16;
    n_out = g * 16 + j;
  } else {
    r0 = wid * 2;
    r1 = r0 + 1;
    n_out = r0;
  }
  if (r0 >= p.N) return;
  const bool has1 = r1 < p.N;
  const int nch = p.K >> 3;  // 16-B chunks per row
  const u32x4_t* w0 = (const u32x4_t*)(p.W + (size_t)r0 * p.K);
  const u32x4_t* w1 = (const u32x4_t*)(p.W + (size_t)(has1 ? r1 : r0) * p.K);
  float acc0[MR], acc1[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) { acc0[m] = 0.f; acc1[m] = 0.f; }
  for (int c = lane; c < nch; c += 64) {
    const u32x4_t a = __builtin_nontemporal_load(w0 + c);
    const u32x4_t b = __builtin_nontemporal_load(w1 + c);
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      const int mm = m < p.M ? m : p.M - 1;
      const u32x4_t xv = *(const u32x4_t*)(p.x + (size_t)mm * p.K + (size_t)c * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x0 = TT::to_f32(xv[e] & 0xffff), x1 = TT::to_f32(xv[e] >> 16);
        acc0[m] += TT::to_f32(a[e] & 0xffff) * x0 + TT::to_f32(a[e] >> 16) * x1;
        acc1[m] += TT::to_f32(b[e] & 0xffff) * x0 + TT::to_f32(b[e] >> 16) * x1;
      }
    }
  }
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    acc0[m] = wave_sum(acc0[m]);
    acc1[m] = wave_sum(acc1[m]);
  }
  if (lane == 0) {
    const int ncols = p.glu ? p.N / 2 : p.N;
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      if (m >= p.M) break;
      float v0, v1 = 0.f;
      if (p.glu) {
        v0 = acc0[m] * apply_act(acc1[m], p.act);
      } else {
        v0 = apply_act(acc0[m], p.act);
        v1 = apply_act(acc1[m], p.act);
      }
      const size_t o = (size_t)m * ncols + n_out;
      if (p.residual) {
        v0 += p.residual[o];
        if (!p.glu && has1) v1 += p.residual[o + 1];
      }
      if (p.out_dtype == SX_F32) {
        ((float*)p.y)[o] = v0;
        if (!p.glu && has1) ((float*)p.y)[o + 1] = v1;
      } else {
        unsigned short* yy = (unsigned short*)p.y;
        yy[o] = p.out_dtype == SX_BF16 ? BF16::from_f32(v0) : F16::from_f32(v0);
        if (!p.glu && has1) yy[o + 1] = p.out_dtype == SX_BF16 ? BF16::from_f32(v1) : F16::from_f32(v1);
      }
    }
  }
}

// ---- skinny GEMM for 2..16 activation rows (lock-step batched decode): y[M][N] = x[M][K] · W[N][K]^T on MFMA -----------------
// The VALU GEMV above costs M FMAs per weight element and turns compute-bound at M = 8 (≈2x its M = 1 time). Here the
// weight rows are the 16x16x32 MFMA's row operand and x^T (M padded to 16 columns) its column operand, so any M <= 16
// costs the same and the kernel stays on the HBM roofline. Block = 4 waves = 4-way split of K; a wave owns R 16-row groups
// (R = 2 = exactly one GLU group [16 linear | 16 gate]) and streams 128 contiguous bytes of each row per k-step (two 16-B
// non-temporal loads per lane; each load instruction covers one 64-B half line of 16 rows: lane group g = lane >> 4 reads
// bytes [16g, 16g + 16) of it, which is exactly the MFMA's k-slot layout, no shuffle). 4 k-steps are in flight per wave
// (8-16 KB). Partial sums meet in LDS; wave 0 runs the fused epilogue (act / GLU / fp32 residual / store).
// TAIL (w_layout 2, R = 2): the workgroup owns 20 weight rows — MFMA row block 0 = rows 0..15, block 1 = rows 16..19 (its other
// 12 operand rows read a zero line). N = 5120 is 320 16-row groups on 256 CUs: the 64 CUs that get two of them set the time
// (o-proj 3.4 TB/s); as 256 groups of 20 every CU streams the same bytes and no cross-workgroup reduction is needed.
__device__ const unsigned g_zero_line[16] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};

// the hi plane of an fp16 pair saturates at the largest finite value (csrc/precise.hip split1 does the same): the remainder travels in lo
__device__ __forceinline__ f32x4_t sat_f16(f32x4_t x) {
  return (f32x4_t){__builtin_amdgcn_fmed3f(x[0], -65504.f, 65504.f), __builtin_amdgcn_fmed3f(x[1], -65504.f, 65504.f),
                   __builtin_amdgcn_fmed3f(x[2], -65504.f, 65504.f), __builtin_amdgcn_fmed3f(x[3], -65504.f, 65504.f)};
}
// lo plane of four fp32 values whose hi plane (two packed dwords) is `hi`: rn16(x - float(hi)), same element type
__device__ __forceinline__ u32x2_t lo_plane(f32x4_t x, u32x2_t hi, bool bf16) {
  u32x2_t lo;
  if (bf16) {
    lo[0] = pack2<BF16>(x[0] - BF16::to_f32(hi[0] & 0xffff), x[1] - BF16::to_f32(hi[0] >> 16));
    lo[1] = pack2<BF16>(x[2] - BF16::to_f32(hi[1] & 0xffff), x[3] - BF16::to_f32(hi[1] >> 16));
  } else {
    lo[0] = pack2<F16>(x[0] - F16::to_f32(hi[0] & 0xffff), x[1] - F16::to_f32(hi[0] >> 16));
    lo[1] = pack2<F16>(x[2] - F16::to_f32(hi[1] & 0xffff), x[3] - F16::to_f32(hi[1] >> 16));
  }
  return lo;
}

// MB = 2 (17..32 activation rows, lock-step batch 32): the x operand is two 16-row blocks — tiles [2][K/32][16][32] or rows 16..31 of
// the row-major x — and every weight fragment feeds two MFMAs, so the weights still stream ONCE for all 32 sequences. Outputs, the
// 16-bit tiled output, x16_out and the sums of squares follow the same block structure (row m = 16 b + r).
template <typename TT, int R, int NWV, int U, bool TAIL = false, int MB = 1>
__global__ __launch_bounds__(NWV * 64) void gemm_skinny_kernel(const GemvP p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef typename TT::vec8 vec8;
  static_assert(!TAIL || R == 2, "the 20-row variant is two MFMA row blocks");
  __shared__ float red[NWV - 1][R * MB][64][4];   // waves 1 .. NWV-1 hand their partial sums to wave 0 (which keeps its own in registers)
  // the wave index as a SCALAR: everything derived from it (k range, round counts) then lives in SGPRs and the guards below
  // are scalar branches. With a VGPR-derived count the compiler predicates the guarded MFMAs through EXEC instead — and
  // MFMA ignores EXEC: the skipped k-steps' (never loaded) registers were multiplied in (found by tools/lab/gemv_lab).
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int n0 = TAIL ? blockIdx.x * 20 : blockIdx.x * 16 * R;
  const int nks = p.K >> 6;
  const int S = gridDim.y;                                           // split-K workgroups per row group (1 = none)
  const int nsl = NWV * S, sl = blockIdx.y * NWV + wave;             // k slices: split-K workgroups x waves
  const int ks0 = sl * nks / nsl, ks1 = (sl + 1) * nks / nsl;
  // x blocks: RB row blocks of 16 activation rows each, times the operand planes — planes2 (fp32-grade activations, x = hi + lo): blocks
  // 0 .. RB-1 = the hi plane's row blocks, RB .. 2 RB - 1 = the lo plane's (MB = 2: 16 rows; MB = 4, round 6: 17..32 rows)
  const int RB = p.planes2 ? MB / 2 : MB;
  bool mvalid[MB];
#pragma unroll
  for (int b = 0; b < MB; ++b) mvalid[b] = r + 16 * (b % RB) < p.M;
  // row-major x: a load instruction touches 16 rows 2K bytes apart — for K = 5120 that stride is a multiple of 16 cache
  // lines, so all 16 rows (and every wave on the chip, in step) hit the same L2 channel. Tiled x: the same instruction reads
  // one contiguous 1-KB tile, consecutive k-steps are consecutive tiles.
  const unsigned short* xp[MB];
#pragma unroll
  for (int b = 0; b < MB; ++b)
    xp[b] = p.x_tiled ? p.x + (size_t)b * (size_t)p.K * 16 + r * 32 + 8 * g : p.x + (size_t)(mvalid[b] ? r + 16 * b : 0) * p.K + 8 * g;
  const size_t xstep = p.x_tiled ? 1024 : 64, xhalf = p.x_tiled ? 512 : 32;
  // row-major W: a load instruction covers one 64-B half line of 16 rows (row stride 2K bytes). Decode layout: the same
  // instruction covers one contiguous 1-KB operand tile, a wave's k range is one contiguous stream.
  const unsigned short* wp[R];
  const size_t kstep = TAIL ? 1280 : (p.packed ? 1024 : 64), khalf = TAIL ? 640 : (p.packed ? 512 : 32);   // elements per 64-wide k-step / to its 2nd half
  size_t kstep_q[R], khalf_q[R];
#pragma unroll
  for (int q = 0; q < R; ++q) {
    kstep_q[q] = kstep; khalf_q[q] = khalf;
    wp[q] = p.packed ? p.W + (size_t)(n0 / 16 + q) * (size_t)(p.K >> 5) * 512 + r * 32 + 8 * g
                     : p.W + (size_t)(n0 + q * 16 + r) * p.K + 8 * g;
  }
  if (TAIL) {   // [N/20][K/32][20 rows][32 k]: a 32-k slab of the group is 1280 B = rows 0..15 (the 1-KB MFMA tile) + rows 16..19
    wp[0] = p.W + (size_t)blockIdx.x * (size_t)(p.K >> 5) * 640 + r * 32 + 8 * g;
    if (r < 4) wp[R - 1] = wp[0] + 512;
    else { wp[R - 1] = (const unsigned short*)g_zero_line; kstep_q[R - 1] = 0; khalf_q[R - 1] = 0; }   // operand rows 4..15 of block 1 = 0
  }
  f32x4_t acc[R][MB];
#pragma unroll
  for (int q = 0; q < R; ++q)
#pragma unroll
    for (int b = 0; b < MB; ++b) acc[q][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  // folded RMSNorm, consumer side: rstd of row r from the producer's per-workgroup sums of squares [16][parts]. Done HERE, ahead of
  // the weight stream, by all 256 threads at once: thread (slice = tid >> 4, r) loads its 1/16 of row r's list as independent 16-B
  // loads (a serial chain of 20 dependent L2 round trips per workgroup cost 5.7 us per launch), two shuffles + one LDS hop add the
  // 16 slices in a fixed order: every workgroup computes the same bits.
  __shared__ float ssq_red[NWV][16 * MB];
  float rstd_fold[MB];
#pragma unroll
  for (int b = 0; b < MB; ++b) rstd_fold[b] = 1.f;
  if (p.ssq_in) {
    const int per = p.ssq_parts >> 4;                                   // parts % 64 == 0 (host-checked) → per % 4 == 0
#pragma unroll
    for (int b = 0; b < MB; ++b) {
      if (b >= RB) break;                                               // the lo plane's blocks are the same rows: no list of their own
      const f32x4_t* src = (const f32x4_t*)(p.ssq_in + (size_t)(r + 16 * b) * p.ssq_parts + (size_t)(wave * 4 + g) * per);
      float s0 = 0.f;
#pragma unroll 5
      for (int i = 0; i < (per >> 2); ++i) {
        const f32x4_t t = src[i];
        s0 += (t[0] + t[1]) + (t[2] + t[3]);
      }
      s0 += __shfl_xor(s0, 16, 64);
      s0 += __shfl_xor(s0, 32, 64);
      if (g == 0) ssq_red[wave][r + 16 * b] = s0;
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < MB; ++b) {
      if (b >= RB) break;
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < NWV; ++w) tot += ssq_red[w][r + 16 * b];
      rstd_fold[b] = __builtin_amdgcn_rsqf(tot * p.ssq_inv_dim + p.ssq_eps);
    }
  }
  // Rounds of U k-steps through TWO register sets: round i+1's loads are issued before round i is consumed, so 1-2 rounds
  // (U..2U KB per row group and wave) are in flight at every moment. The pipelined loop has no branch inside (the waitcnt
  // pass then emits exact vmcnt(N) waits; with a guard inside it falls back to vmcnt(0) before the first MFMA). Last full
  // rounds and the < U remainder are peeled. Measured (tools/lab/gemv_lab, profiles/r3_ab_experiments.md §6): the deeper
  // flight alone changes nothing (+-1 %) — the wide shapes already stream at 6.5-6.9 TB/s net of the 3-4 us launch cost;
  // what moved the numbers was the x layout (L2 channel conflicts) and, for K = 13824, split-K.
  struct Frag { u32x4_t wa[U][R], wb[U][R], xa[U][MB], xb[U][MB]; };
  auto load_round = [&](Frag& f, int ks, int cnt) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (u < cnt) {
        const size_t k = (size_t)(ks + u) * xstep;
#pragma unroll
        for (int q = 0; q < R; ++q) {
          f.wa[u][q] = __builtin_nontemporal_load((const u32x4_t*)(wp[q] + (size_t)(ks + u) * kstep_q[q]));
          f.wb[u][q] = __builtin_nontemporal_load((const u32x4_t*)(wp[q] + (size_t)(ks + u) * kstep_q[q] + khalf_q[q]));
        }
#pragma unroll
        for (int b = 0; b < MB; ++b) {
          f.xa[u][b] = *(const u32x4_t*)(xp[b] + k);
          f.xb[u][b] = *(const u32x4_t*)(xp[b] + k + xhalf);
        }
      }
    }
  };
  auto consume = [&](const Frag& f, int cnt) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (u < cnt) {
        // lanes of rows >= M carry row 0's x: their output columns (m >= M) are never stored, and an MFMA output column depends
        // on its own B column only — no select on the loaded registers (a VALU op on them at the top of the round makes the
        // waitcnt pass drain every load in flight)
#pragma unroll
        for (int q = 0; q < R; ++q)
#pragma unroll
          for (int b = 0; b < MB; ++b) {
            acc[q][b] = TT::mfma16(__builtin_bit_cast(vec8, f.wa[u][q]), __builtin_bit_cast(vec8, f.xa[u][b]), acc[q][b]);
            acc[q][b] = TT::mfma16(__builtin_bit_cast(vec8, f.wb[u][q]), __builtin_bit_cast(vec8, f.xb[u][b]), acc[q][b]);
          }
      }
    }
  };
  {
    Frag A, B;
    const int full = (ks1 - ks0) / U, rem = (ks1 - ks0) % U;
    int i = 0;
    if (full > 0) {
      load_round(A, ks0, U);
      for (; i + 2 < full; i += 2) {
        // sched_barrier: the machine scheduler otherwise hoists the next round's loads above this round's MFMAs (renaming
        // the registers), which turns the loop back into "issue everything, then wait for everything"
        load_round(B, ks0 + (i + 1) * U, U);
        __builtin_amdgcn_sched_barrier(0);
        consume(A, U);
        __builtin_amdgcn_sched_barrier(0);
        load_round(A, ks0 + (i + 2) * U, U);
        __builtin_amdgcn_sched_barrier(0);
        consume(B, U);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (i + 2 == full) {
        load_round(B, ks0 + (i + 1) * U, U);
        consume(A, U);
        if (rem) load_round(A, ks0 + full * U, rem);
        consume(B, U);
      } else {
        if (rem) load_round(B, ks0 + full * U, rem);
        consume(A, U);
        if (rem) { consume(B, rem); }
      }
      if (i + 2 == full && rem) consume(A, rem);
    } else if (rem) {
      load_round(A, ks0, rem);
      consume(A, rem);
    }
  }
  if (wave != 0) {
#pragma unroll
    for (int q = 0; q < R; ++q)
#pragma unroll
      for (int b = 0; b < MB; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wave - 1][q * MB + b][lane][e] = acc[q][b][e];
  }
  __syncthreads();
  if (wave != 0) return;
  // lane holds y[m = 16 b + r][n0 + q*16 + 4g + e], e = 0..3; the waves' partial sums are added in wave order (0 + w0 == w0: the same bits as
  // when wave 0 went through LDS too)
  f32x4_t v[R][MB];
#pragma unroll
  for (int q = 0; q < R; ++q)
#pragma unroll
    for (int b = 0; b < MB; ++b)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = acc[q][b][e];
#pragma unroll
        for (int w = 0; w + 1 < NWV; ++w) t += red[w][q * MB + b][lane][e];
        v[q][b][e] = t;
      }
  if (S > 1) {
    // split-K: every workgroup publishes its partial [16][16R] block; the LAST one to arrive (per row group) adds the S partials
    // in split order — the result does not depend on which workgroup that is — and runs the epilogue. The counter is left at 0.
#pragma unroll
    for (int q = 0; q < R; ++q)
#pragma unroll
      for (int b = 0; b < MB; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (!(TAIL && q == 1 && g != 0))
            __hip_atomic_store(p.ws_part + ((size_t)blockIdx.y * (16 * MB) + 16 * b + r) * p.N + n0 + q * 16 + 4 * g + e, v[q][b][e],
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // No cache-wide fence (an agent-scope release / acquire writes back and invalidates the whole per-XCD L2 — measured 4x
    // slower kernels): the partials themselves move as agent-scope (sc1) atomics, which are performed at the device's
    // coherence point, so ordering them against the counter only needs "my stores are acknowledged" before the count and
    // "count seen" before the loads.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(p.ws_cnt + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    old = __shfl(old, 0);
    if (old != (unsigned)(S - 1)) return;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < R; ++q)
#pragma unroll
      for (int b = 0; b < MB; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = 0.f;
          if (!(TAIL && q == 1 && g != 0))
            for (int sidx = 0; sidx < S; ++sidx)
              t += __hip_atomic_load(p.ws_part + ((size_t)sidx * (16 * MB) + 16 * b + r) * p.N + n0 + q * 16 + 4 * g + e, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT);
          v[q][b][e] = t;
        }
    if (lane == 0) __hip_atomic_store(p.ws_cnt + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (MB >= 2 && p.planes2) {   // hi-plane + lo-plane products of the same rows: row block rb = blocks rb and RB + rb
#pragma unroll
    for (int q = 0; q < R; ++q)
#pragma unroll
      for (int b = 0; b < MB / 2; ++b) v[q][b] += v[q][MB / 2 + b];
  }
  if (p.ssq_in) {
#pragma unroll
    for (int q = 0; q < R; ++q)
#pragma unroll
      for (int b = 0; b < MB; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (b < RB) v[q][b][e] *= rstd_fold[b];
  }
  const int ncols = p.glu ? p.N / 2 : p.N;
  const size_t lo_off = (size_t)ncols * 16 * (size_t)((p.M + 15) >> 4);   // planes out: the lo plane follows the hi plane's row blocks
#pragma unroll
  for (int b = 0; b < MB; ++b) {
    if (b >= RB) break;
    float ssq_l = 0.f;
    const int m = r + 16 * b;                               // activation row; 16-bit tiles: block b of [RB][ncols/32][16][32]
    const size_t tblk = (size_t)b * (size_t)ncols * 16;
#pragma unroll
    for (int q = 0; q < R; ++q) {
      if (!mvalid[b]) break;
      if (TAIL && q == 1 && g != 0) break;           // block 1 holds rows 16..19 only: output columns n0 + 16 .. n0 + 19 (lane group 0)
      f32x4_t o;
      int col;
      if (p.glu) {
        if (2 * q + 1 >= R) break;                       // GLU: row groups (2 q, 2 q + 1) = one packed [16 linear | 16 gate] group
        col = (blockIdx.x * (R / 2) + q) * 16 + 4 * g;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = v[2 * q][b][e] * apply_act(v[(2 * q + 1) % R][b][e], p.act);
      } else {
        col = n0 + q * 16 + 4 * g;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = apply_act(v[q][b][e], p.act);
      }
      const size_t off = (size_t)m * ncols + col;
      if (p.residual) o += *(const f32x4_t*)(p.residual + off);
      if (p.y_tiled) {      // 16-bit operand tiles [ncols/32][16][32] for the next skinny GEMM (x_layout = 1)
        u32x2_t w2;
        if (p.out_dtype == SX_BF16) { w2[0] = pack2<BF16>(o[0], o[1]); w2[1] = pack2<BF16>(o[2], o[3]); }
        else if (p.out_planes) { const f32x4_t oc = sat_f16(o); w2[0] = pack2<F16>(oc[0], oc[1]); w2[1] = pack2<F16>(oc[2], oc[3]); }   // hi saturates, lo carries the rest
        else { w2[0] = pack2<F16>(o[0], o[1]); w2[1] = pack2<F16>(o[2], o[3]); }
        unsigned short* yt = (unsigned short*)p.y + tblk + (size_t)(col >> 5) * 512 + (size_t)r * 32 + (col & 31);
        *(u32x2_t*)yt = w2;
        if (p.out_planes) *(u32x2_t*)(yt + lo_off) = lo_plane(o, w2, p.out_dtype == SX_BF16);   // lo plane = o - hi
      } else if (p.out_dtype == SX_F32) {
        *(f32x4_t*)((float*)p.y + off) = o;
        if (p.x16_out) {     // folded RMSNorm, producer side: the new residual stream also as the next GEMV's 16-bit operand tiles
          f32x4_t og = o;      // precise mode: gamma of the NEXT norm goes onto the activation (weights stay exact), two planes
          if (p.x16_gamma) og = o * *(const f32x4_t*)(p.x16_gamma + col);
          u32x2_t w2;
          if (std::is_same<TT, BF16>::value) { w2[0] = pack2<BF16>(og[0], og[1]); w2[1] = pack2<BF16>(og[2], og[3]); }
          else if (p.out_planes) { const f32x4_t oc = sat_f16(og); w2[0] = pack2<F16>(oc[0], oc[1]); w2[1] = pack2<F16>(oc[2], oc[3]); }
          else { w2[0] = pack2<F16>(og[0], og[1]); w2[1] = pack2<F16>(og[2], og[3]); }
          unsigned short* xt = p.x16_out + tblk + (size_t)(col >> 5) * 512 + (size_t)r * 32 + (col & 31);
          *(u32x2_t*)xt = w2;
          if (p.out_planes) *(u32x2_t*)(xt + lo_off) = lo_plane(og, w2, std::is_same<TT, BF16>::value);
          ssq_l += o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3];
        }
      } else {
        u32x2_t w2;
        if (p.out_dtype == SX_BF16) { w2[0] = pack2<BF16>(o[0], o[1]); w2[1] = pack2<BF16>(o[2], o[3]); }
        else { w2[0] = pack2<F16>(o[0], o[1]); w2[1] = pack2<F16>(o[2], o[3]); }
        *(u32x2_t*)((unsigned short*)p.y + off) = w2;
      }
    }
    if (p.ssq_out) {                                     // (whole wave: rows >= M contribute zeros and are never read)
      ssq_l += __shfl_xor(ssq_l, 16, 64);
      ssq_l += __shfl_xor(ssq_l, 32, 64);
      if (g == 0) p.ssq_out[(size_t)m * gridDim.x + blockIdx.x] = ssq_l;      // [16 MB][parts]
    }
  }
#endif
}

// ---- decode attention: grid (H, nsplit); 16-lane groups own one key row per iteration ---------------------------
template <typename TT>
__global__ __launch_bounds__(256) void attn_decode_kernel(const unsigned short* q, const unsigned short* kc,
                                                          const unsigned short* vc, float* scratch,
                                                          const int* ctx_len_dev, int D, int Tmax, int nsplit,
                                                          float scale, long long seq_stride, long long q_stride) {
  __shared__ float red[16][132];  // 16 lane-groups x (D<=128 outputs + m + l)
  const int h = blockIdx.x, sp = blockIdx.y, g = blockIdx.z, H = gridDim.x;
  const int ctx = ctx_len_dev[g];
  q += (size_t)g * q_stride;
  kc += (size_t)g * seq_stride;
  vc += (size_t)g * seq_stride;
  scratch += (size_t)g * H * nsplit * (D + 2);
  const int chunk = (ctx + nsplit - 1) / nsplit;
  const int t0 = sp * chunk, t1 = min(ctx, t0 + chunk);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = wave * 4 + (lane >> 4);  // 0..15
  const int dl = lane & 15;                // 16-B chunk of the head dim (8 elements); D = 128 → 16 chunks
  const bool dvalid = dl * 8 < D;
  float qv[8];
  {
    u32x4_t raw = {0u, 0u, 0u, 0u};
    if (dvalid) raw = *(const u32x4_t*)(q + (size_t)h * D + dl * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      qv[2 * e] = TT::to_f32(raw[e] & 0xffff) * scale;
      qv[2 * e + 1] = TT::to_f32(raw[e] >> 16) * scale;
    }
  }
  float m_run = -INFINITY, l_run = 0.f, o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = 0.f;
  const unsigned short* kh = kc + (size_t)h * Tmax * D;
  const unsigned short* vh = vc + (size_t)h * Tmax * D;
  for (int t = t0 + grp; t < t1; t += 16) {
    u32x4_t kr = {0u, 0u, 0u, 0u}, vr = {0u, 0u, 0u, 0u};
    if (dvalid) {
      kr = *(const u32x4_t*)(kh + (size_t)t * D + dl * 8);
      vr = *(const u32x4_t*)(vh + (size_t)t * D + dl * 8);
    }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      s += TT::to_f32(kr[e] & 0xffff) * qv[2 * e] + TT::to_f32(kr[e] >> 16) * qv[2 * e + 1];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    s += __shfl_xor(s, 8, 64);
    const float m_new = fmaxf(m_run, s);
    const float alpha = __expf(m_run - m_new);  // exp(-inf) = 0 on the first key
    const float pr = __expf(s - m_new);
    l_run = l_run * alpha + pr;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[2 * e] = o[2 * e] * alpha + pr * TT::to_f32(vr[e] & 0xffff);
      o[2 * e + 1] = o[2 * e + 1] * alpha + pr * TT::to_f32(vr[e] >> 16);
    }
    m_run = m_new;
  }
  if (dvalid) {
#pragma unroll
    for (int e = 0; e < 8; ++e) red[grp][dl * 8 + e] = o[e];
  }
  if (dl == 0) {
    red[grp][128] = m_run;
    red[grp][129] = l_run;
  }
  __syncthreads();
  // combine the 16 groups: thread d < D
  const int d = threadIdx.x;
  float* out = scratch + ((size_t)h * nsplit + sp) * (D + 2);
  float mg = -INFINITY;
#pragma unroll
  for (int g = 0; g < 16; ++g) mg = fmaxf(mg, red[g][128]);
  if (d < D) {
    float acc = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const float mgk = red[g][128];
      const float w = (mgk == -INFINITY) ? 0.f : __expf(mgk - mg);
      acc += w * red[g][d];
    }
    out[d] = acc;
  }
  if (d == 0) {
    float lsum = 0.f;
    for (int g = 0; g < 16; ++g) {
      const float mgk = red[g][128];
      lsum += (mgk == -INFINITY) ? 0.f : __expf(mgk - mg) * red[g][129];
    }
    out[D] = mg;
    out[D + 1] = lsum;
  }
}

// ---- fused decode attention: RoPE of the new token + KV append + split-KV attention + combine, ONE launch ------------------------
// The three-kernel form above costs 6 us (rope) + 18 us (attention) + 5.5 us (combine) per layer, two of them launch-latency
// sized. Here every (head, split, sequence) workgroup rotates q itself (128 values), the one 16-lane group whose key index is the
// new position takes K / V of the new token straight from the qkv row (rotating K, writing both into the cache for later
// steps), and the LAST split of a head to arrive (agent-scope partial stores + arrival counter, as in the skinny GEMM's
// split-K) combines the partials in split order. Same arithmetic, same order, same roundings as rope_kv_append_kernel +
// attn_decode_kernel + attn_decode_combine_kernel: bit-identical output and cache contents.
struct AttnDecP {
  const unsigned short* qkv;   // [G][3*H*D]: q | k | v of the new token, un-rotated
  unsigned short* kc;
  unsigned short* vc;
  unsigned short* out;
  float* scratch;              // [G][H][nsplit][D + 2]
  unsigned* counters;          // [G*H], zero between launches
  const float* cos_t;
  const float* sin_t;
  const int* pos_dev;          // [G]
  int H, D, Tmax, nsplit, tiled;
  float scale;
  long long seq_stride;
};

template <typename TT>
__device__ __forceinline__ void rope_chunk(const unsigned short* base, int dl, int D, const float* cos_t, const float* sin_t, int pos,
                                           float* out8) {
  // rotated elements 8dl .. 8dl+7 of a D-wide head vector at `base`, rounded to the activation dtype (rope_kv_append_kernel)
  const int half = D / 2, j0 = dl * 8;
  const bool lo = j0 < half;
  const u32x4_t a = *(const u32x4_t*)(base + j0);
  const u32x4_t b = *(const u32x4_t*)(base + (lo ? j0 + half : j0 - half));
  // the 8 table entries of this chunk are contiguous (half % 8 == 0): two 16-B loads per table
  const size_t tb = (size_t)pos * half + (lo ? j0 : j0 - half);
  const f32x4_t c0 = *(const f32x4_t*)(cos_t + tb), c1 = *(const f32x4_t*)(cos_t + tb + 4);
  const f32x4_t s0 = *(const f32x4_t*)(sin_t + tb), s1 = *(const f32x4_t*)(sin_t + tb + 4);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float c = TT::to_f32(TT::from_f32(e < 4 ? c0[e & 3] : c1[e & 3]));
    const float sn = TT::to_f32(TT::from_f32(e < 4 ? s0[e & 3] : s1[e & 3]));
    const float x = TT::to_f32((e & 1) ? (a[e >> 1] >> 16) : (a[e >> 1] & 0xffff));
    const float y = TT::to_f32((e & 1) ? (b[e >> 1] >> 16) : (b[e >> 1] & 0xffff));
    // first half: x*c - y*s (y = partner in the second half); second half: x*c + y*s (y = partner in the first half)
    out8[e] = TT::to_f32(TT::from_f32(lo ? x * c - y * sn : x * c + y * sn));
  }
}

template <typename TT>
__global__ __launch_bounds__(256) void attn_decode_fused_kernel(const AttnDecP p) {
  __shared__ float red[16][132];
  __shared__ unsigned s_last;
  const int h = blockIdx.x, sp = blockIdx.y, g = blockIdx.z, H = p.H, D = p.D, nsplit = p.nsplit;
  const int pos = p.pos_dev[g];
  const bool pos_ok = pos >= 0 && pos < p.Tmax;
  const int ctx = pos_ok ? pos + 1 : (pos < 0 ? 0 : p.Tmax);   // a position past the cache: attend to the cache only, write nothing
  const unsigned short* row = p.qkv + (size_t)g * 3 * H * D;
  unsigned short* kh = p.kc + (size_t)g * p.seq_stride + (size_t)h * p.Tmax * D;
  unsigned short* vh = p.vc + (size_t)g * p.seq_stride + (size_t)h * p.Tmax * D;
  float* scratch = p.scratch + (size_t)g * H * nsplit * (D + 2);
  const int chunk = (ctx + nsplit - 1) / nsplit;
  const int t0 = sp * chunk, t1 = min(ctx, t0 + chunk);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = wave * 4 + (lane >> 4);
  const int dl = lane & 15;
  const bool dvalid = dl * 8 < D;
  const int rpos = pos_ok ? pos : 0;
  float qv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) qv[e] = 0.f;
  if (dvalid) {
    rope_chunk<TT>(row + (size_t)h * D, dl, D, p.cos_t, p.sin_t, rpos, qv);
#pragma unroll
    for (int e = 0; e < 8; ++e) qv[e] *= p.scale;
  }
  float m_run = -INFINITY, l_run = 0.f, o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = 0.f;
  for (int t = t0 + grp; t < t1; t += 16) {
    float kf[8], vf[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { kf[e] = 0.f; vf[e] = 0.f; }
    if (dvalid) {
      if (pos_ok && t == pos) {      // the new token: from the qkv row, and into the cache
        rope_chunk<TT>(row + (size_t)(H + h) * D, dl, D, p.cos_t, p.sin_t, pos, kf);
        const u32x4_t vr = *(const u32x4_t*)(row + (size_t)(2 * H + h) * D + dl * 8);
        u32x4_t kw;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          kw[e] = (unsigned)TT::from_f32(kf[2 * e]) | ((unsigned)TT::from_f32(kf[2 * e + 1]) << 16);
          vf[2 * e] = TT::to_f32(vr[e] & 0xffff);
          vf[2 * e + 1] = TT::to_f32(vr[e] >> 16);
        }
        *(u32x4_t*)(kh + (size_t)pos * D + dl * 8) = kw;
        *(u32x4_t*)(vh + (size_t)pos * D + dl * 8) = vr;
      } else {
        const u32x4_t kr = *(const u32x4_t*)(kh + (size_t)t * D + dl * 8);
        const u32x4_t vr = *(const u32x4_t*)(vh + (size_t)t * D + dl * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          kf[2 * e] = TT::to_f32(kr[e] & 0xffff); kf[2 * e + 1] = TT::to_f32(kr[e] >> 16);
          vf[2 * e] = TT::to_f32(vr[e] & 0xffff); vf[2 * e + 1] = TT::to_f32(vr[e] >> 16);
        }
      }
    }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) s += kf[2 * e] * qv[2 * e] + kf[2 * e + 1] * qv[2 * e + 1];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    s += __shfl_xor(s, 8, 64);
    const float m_new = fmaxf(m_run, s);
    const float alpha = __expf(m_run - m_new);
    const float pr = __expf(s - m_new);
    l_run = l_run * alpha + pr;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = o[e] * alpha + pr * vf[e];
    m_run = m_new;
  }
  if (dvalid) {
#pragma unroll
    for (int e = 0; e < 8; ++e) red[grp][dl * 8 + e] = o[e];
  }
  if (dl == 0) {
    red[grp][128] = m_run;
    red[grp][129] = l_run;
  }
  __syncthreads();
  const int d = threadIdx.x;
  float* part = scratch + ((size_t)h * nsplit + sp) * (D + 2);
  float mg = -INFINITY;
#pragma unroll
  for (int gg = 0; gg < 16; ++gg) mg = fmaxf(mg, red[gg][128]);
  if (nsplit == 1) {
    // one split per head: this workgroup holds the whole result — no partials in memory, no counter, no second pass. The
    // split's (acc, m, l) go through LDS and the combine's arithmetic runs on them unchanged (same expressions as the split
    // path and the combine below, so the bits are those of the three-launch form)
    __shared__ float fin[132];
    if (d < D) {
      float acc = 0.f;
#pragma unroll
      for (int gg = 0; gg < 16; ++gg) {
        const float mgk = red[gg][128];
        const float w = (mgk == -INFINITY) ? 0.f : __expf(mgk - mg);
        acc += w * red[gg][d];
      }
      fin[d] = acc;
    }
    if (d == 0) {
      float lsum = 0.f;
      for (int gg = 0; gg < 16; ++gg) {
        const float mgk = red[gg][128];
        lsum += (mgk == -INFINITY) ? 0.f : __expf(mgk - mg) * red[gg][129];
      }
      fin[D] = mg;
      fin[D + 1] = lsum;
    }
    __syncthreads();
    if (d >= D) return;
    float mall = -INFINITY;
    mall = fmaxf(mall, fin[D]);
    const float ms = fin[D];
    const float w1 = (ms == -INFINITY) ? 0.f : __expf(ms - mall);
    float a2 = 0.f, l2 = 0.f;
    a2 += w1 * fin[d];
    l2 += w1 * fin[D + 1];
    const int col1 = h * D + d;
    const size_t o1 = p.tiled ? (size_t)(g >> 4) * (size_t)H * D * 16 + (size_t)(col1 >> 5) * 512 + (size_t)(g & 15) * 32 + (col1 & 31)
                              : (size_t)g * H * D + col1;
    p.out[o1] = TT::from_f32(l2 > 0.f ? a2 / l2 : 0.f);
    return;
  }
  if (d < D) {
    float acc = 0.f;
#pragma unroll
    for (int gg = 0; gg < 16; ++gg) {
      const float mgk = red[gg][128];
      const float w = (mgk == -INFINITY) ? 0.f : __expf(mgk - mg);
      acc += w * red[gg][d];
    }
    __hip_atomic_store(part + d, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (d == 0) {
    float lsum = 0.f;
    for (int gg = 0; gg < 16; ++gg) {
      const float mgk = red[gg][128];
      lsum += (mgk == -INFINITY) ? 0.f : __expf(mgk - mg) * red[gg][129];
    }
    __hip_atomic_store(part + D, mg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(part + D + 1, lsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // arrival: every thread's partial stores are acknowledged, then one count per workgroup; the last split combines
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0)
    s_last = __hip_atomic_fetch_add(p.counters + (size_t)g * H + h, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(nsplit - 1);
  __syncthreads();
  if (!s_last) return;
  if (threadIdx.x == 0) __hip_atomic_store(p.counters + (size_t)g * H + h, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (d >= D) return;
  const float* base = scratch + (size_t)h * nsplit * (D + 2);
  float mall = -INFINITY, acc = 0.f, l = 0.f;
  if (nsplit <= 8) {
    // all 3 x nsplit partial loads in flight together (they miss to memory: the partials came from other CUs / XCDs), then the
    // same arithmetic in the same order as the loop below
    float pm[8], pl[8], pa[8];
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2) {
      const size_t o2 = (size_t)(s2 < nsplit ? s2 : 0) * (D + 2);
      pm[s2] = __hip_atomic_load(base + o2 + D, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pl[s2] = __hip_atomic_load(base + o2 + D + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pa[s2] = __hip_atomic_load(base + o2 + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2)
      if (s2 < nsplit) mall = fmaxf(mall, pm[s2]);
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2)
      if (s2 < nsplit) {
        const float w = (pm[s2] == -INFINITY) ? 0.f : __expf(pm[s2] - mall);
        acc += w * pa[s2];
        l += w * pl[s2];
      }
  } else {
    for (int s2 = 0; s2 < nsplit; ++s2)
      mall = fmaxf(mall, __hip_atomic_load(base + (size_t)s2 * (D + 2) + D, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    for (int s2 = 0; s2 < nsplit; ++s2) {
      const float ms = __hip_atomic_load(base + (size_t)s2 * (D + 2) + D, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float w = (ms == -INFINITY) ? 0.f : __expf(ms - mall);
      acc += w * __hip_atomic_load(base + (size_t)s2 * (D + 2) + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      l += w * __hip_atomic_load(base + (size_t)s2 * (D + 2) + D + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  const int col = h * D + d;
  const size_t oo = p.tiled ? (size_t)(g >> 4) * (size_t)H * D * 16 + (size_t)(col >> 5) * 512 + (size_t)(g & 15) * 32 + (col & 31)
                            : (size_t)g * H * D + col;
  p.out[oo] = TT::from_f32(l > 0.f ? acc / l : 0.f);
}

template <typename TT>
__global__ void attn_decode_combine_kernel(const float* scratch, unsigned short* out, int D, int nsplit, int tiled) {
  const int h = blockIdx.x, d = threadIdx.x, g = blockIdx.y, H = gridDim.x;
  if (d >= D) return;
  scratch += (size_t)g * H * nsplit * (D + 2);
  const float* base = scratch + (size_t)h * nsplit * (D + 2);
  float mg = -INFINITY;
  for (int s = 0; s < nsplit; ++s) mg = fmaxf(mg, base[(size_t)s * (D + 2) + D]);
  float acc = 0.f, l = 0.f;
  for (int s = 0; s < nsplit; ++s) {
    const float ms = base[(size_t)s * (D + 2) + D];
    const float w = (ms == -INFINITY) ? 0.f : __expf(ms - mg);
    acc += w * base[(size_t)s * (D + 2) + d];
    l += w * base[(size_t)s * (D + 2) + D + 1];
  }
  const int col = h * D + d;                     // of row g in [G][H*D]; SX_TILED16: tile col/32, row g, column col%32
  const size_t o = tiled ? (size_t)(g >> 4) * (size_t)H * D * 16 + (size_t)(col >> 5) * 512 + (size_t)(g & 15) * 32 + (col & 31)
                         : (size_t)g * H * D + col;      // sequences 16..31: a second block of tiles behind the first
  out[o] = TT::from_f32(l > 0.f ? acc / l : 0.f);
}

// ---- RoPE + KV append ----------------------------------------------------------------------------------------------
template <typename TT>
__global__ void rope_kv_append_kernel(unsigned short* qkv, unsigned short* kc, unsigned short* vc, const float* cos_t,
                                      const float* sin_t, const int* pos0_dev, int T, int H, int D, int Tmax, int G,
                                      long long seq_stride) {
  const int half = D / 2;
  const int64_t total = (int64_t)G * T * H * half;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % half);
    const int h = (int)((i / half) % H);
    const int r = (int)(i / ((int64_t)half * H));     // row of qkv = g*T + t
    const int g = r / T, t = r - g * T;
    int pos = pos0_dev[g] + t;
    const bool pos_ok = pos >= 0 && pos < Tmax;
    if (!pos_ok) pos = 0;                   // keep the table reads in range; the cache write below is skipped
    // tables are rounded to the activation dtype before use (modeling_llama_xformer.py:128-131)
    const float c = TT::to_f32(TT::from_f32(cos_t[(size_t)pos * half + j]));
    const float s = TT::to_f32(TT::from_f32(sin_t[(size_t)pos * half + j]));
    unsigned short* row = qkv + (size_t)r * 3 * H * D;
    unsigned short* qh = row + (size_t)h * D;
    const unsigned short* kh = row + (size_t)(H + h) * D;
    const unsigned short* vh = row + (size_t)(2 * H + h) * D;
    const float q1 = TT::to_f32(qh[j]), q2 = TT::to_f32(qh[j + half]);
    qh[j] = TT::from_f32(q1 * c - q2 * s);
    qh[j + half] = TT::from_f32(q2 * c + q1 * s);
    const float k1 = TT::to_f32(kh[j]), k2 = TT::to_f32(kh[j + half]);
    if (!pos_ok) continue;                  // device-resident position past the cache: never write outside it
    unsigned short* kd = kc + (size_t)g * seq_stride + ((size_t)h * Tmax + pos) * D;
    unsigned short* vd = vc + (size_t)g * seq_stride + ((size_t)h * Tmax + pos) * D;
    kd[j] = TT::from_f32(k1 * c - k2 * s);
    kd[j + half] = TT::from_f32(k2 * c + k1 * s);
    vd[j] = vh[j];
    vd[j + half] = vh[j + half];
  }
}

template <typename TT>
__global__ void embedding_kernel(const int* ids, const unsigned short* table, float* out, int T, int dim) {
  const int64_t total = (int64_t)T * dim;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(i / dim), d = (int)(i % dim);
    out[i] = TT::to_f32(table[(size_t)ids[t] * dim + d]);
  }
}

__global__ void scatter_rows_kernel(const float* src, const int* rows, float* dst, int n, int dim) {
  const int64_t total = (int64_t)n * dim;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / dim), d = (int)(i % dim);
    dst[(size_t)rows[r] * dim + d] = src[i];
  }
}

// dst[(g*seq_rows + step[g])][:] = src[g][:]  (per-sequence hidden-state log of the lock-step batched decode)
__global__ void scatter_rows_step_kernel(const float* src, const int* step, float* dst, int G, int dim, int seq_rows) {
  const int64_t total = (int64_t)G * dim;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int g = (int)(i / dim), d = (int)(i % dim);
    const int st = step[g];
    if (st < 0 || st >= seq_rows) continue;  // device-resident step past the log: drop the row instead of corrupting HBM
    dst[((size_t)g * seq_rows + st) * dim + d] = src[i];
  }
}

// ---- greedy next token with the AutoImageTokenGenerationProcessor rule (generation.py:19-31) ------------------------
__global__ __launch_bounds__(1024) void greedy_next_kernel(float* logits, int vocab, const int* img_ids, int n_img,
                                                           const int* prev_id, int* next_id, int* out_ids,
                                                           const int* step_dev, int ld_logits, int ld_out) {
  // one block per sequence
  logits += (size_t)blockIdx.x * ld_logits;
  prev_id += blockIdx.x;
  next_id += blockIdx.x;
  if (out_ids) out_ids += (size_t)blockIdx.x * ld_out;
  if (step_dev) step_dev += blockIdx.x;
  __shared__ float smax[16];
  __shared__ int sidx[16];
  __shared__ int forced;
  const int prev = *prev_id;
  if (threadIdx.x == 0) forced = -1;
  __syncthreads();
  // prev in img_ids[:-1] → force the next id of the chain (scores[next] = max + 10 in the reference)
  if ((int)threadIdx.x < n_img - 1 && img_ids[threadIdx.x] == prev) atomicMax(&forced, (int)threadIdx.x);
  __syncthreads();
  int result;
  if (forced >= 0) {
    // list.index() returns the FIRST match; ids are unique so max == first
    result = img_ids[forced + 1];
  } else {
    if ((int)threadIdx.x < n_img - 1) logits[img_ids[1 + threadIdx.x]] = 0.0f;
    __syncthreads();
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < vocab; i += 1024) {
      const float v = logits[i];
      if (v > best || (v == best && i < bi)) { best = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { smax[threadIdx.x >> 6] = best; sidx[threadIdx.x >> 6] = bi; }
    __syncthreads();
    best = smax[0];
    bi = sidx[0];
    for (int w = 1; w < 16; ++w)
      if (smax[w] > best || (smax[w] == best && sidx[w] < bi)) { best = smax[w]; bi = sidx[w]; }
    result = bi;
  }
  if (threadIdx.x == 0) {
    *next_id = result;
    if (out_ids) {
      const int st = *step_dev;
      if (st >= 0 && (ld_out <= 0 || st < ld_out)) out_ids[st] = result;   // ld_out doubles as the row capacity
    }
  }
}

inline dim3 gs_grid(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return dim3((unsigned)b);
}

}  // namespace sxk_decode
using namespace sxk_decode;

#define ST ((hipStream_t)stream)
static int g_skinny_var[4] = {0, 0, 0, 1};   // [3] = 64-row workgroups (R = 4): 0 never, 1 where >= 200 workgroups remain (round 6), 2 wherever legal  // tuning hook (sx_gemv_tune): [2] = split-K factor (0 auto, -1 never, 2 / 4 / 8 forced)
extern "C" int sx_gemv_tune(int key, int value) {
  SX_CHECK((key == 2 && (value == -1 || value == 0 || value == 1 || value == 2 || value == 4 || value == 8)) || (key == 1 && (value == 0 || value == 1 || value == 2)) ||
               (key == 3 && value >= 0 && value <= 2),
           "sx_gemv_tune: key %d value %d", key, value);
  g_skinny_var[key] = value;
  return SX_OK;
}
static int g_force_valu_gemv = 0;   // test hook (sx_gemv_force_valu): compare the two GEMV paths
extern "C" int sx_gemv_force_valu(int on) { g_force_valu_gemv = on; return SX_OK; }   // 1 = VALU only, 2 = MFMA whenever legal

// workgroups in x of the MFMA path (must mirror the r2 / gx choice in sx_gemv below)
extern "C" int sx_gemv_ssq_parts(int N, int glu, int w_layout) {
  if (w_layout == 2) return N / 20;
  return (glu || N / 32 >= 256) ? N / 32 : N / 16;
}

extern "C" int sx_gemv(const sx_gemv_args* a, void* stream) {
  SX_CHECK(a && a->x && a->W && a->y, "sx_gemv: null pointer");
  SX_CHECK(a->dtype == SX_F16 || a->dtype == SX_BF16, "sx_gemv: dtype");
  SX_CHECK(a->M >= 1 && a->M <= 32, "sx_gemv: M=%d must be 1..32", a->M);
  SX_CHECK(a->K % 8 == 0 && a->N > 0, "sx_gemv: K %% 8");
  SX_CHECK(!a->glu || a->N % 32 == 0, "sx_gemv: glu needs N %% 32 == 0");
  GemvP p;
  p.x = (const unsigned short*)a->x; p.W = (const unsigned short*)a->W; p.y = a->y; p.residual = a->residual;
  p.M = a->M; p.N = a->N; p.K = a->K; p.out_dtype = a->out_dtype & 0xff; p.act = a->act; p.glu = a->glu;
  p.y_tiled = (a->out_dtype & SX_TILED16) ? 1 : 0;
  SX_CHECK(!p.y_tiled || ((p.out_dtype == SX_F16 || p.out_dtype == SX_BF16) && (a->glu ? a->N / 2 : a->N) % 32 == 0),
           "sx_gemv: SX_TILED16 needs a 16-bit output with n_out %% 32 == 0");
  p.packed = a->w_layout == 1 ? 1 : 0;
  p.ws_cnt = nullptr; p.ws_part = nullptr;
  p.x16_out = (unsigned short*)a->x16_out; p.ssq_out = a->row_ssq_out; p.ssq_in = a->row_ssq_in;
  p.ssq_parts = a->ssq_in_parts; p.ssq_inv_dim = a->ssq_dim > 0 ? 1.0f / (float)a->ssq_dim : 0.f; p.ssq_eps = a->ssq_eps;
  SX_CHECK(!a->row_ssq_in || (a->ssq_in_parts > 0 && a->ssq_in_parts % 64 == 0 && a->ssq_dim > 0 && ((uintptr_t)a->row_ssq_in & 15) == 0),
           "sx_gemv: row_ssq_in needs ssq_in_parts (a multiple of 64), ssq_dim and a 16-B aligned list");
  SX_CHECK((a->x16_out == nullptr) == (a->row_ssq_out == nullptr), "sx_gemv: x16_out and row_ssq_out come together");
  SX_CHECK(!a->x16_out || (p.out_dtype == SX_F32 && !a->glu && a->N % 32 == 0 && !p.y_tiled),
           "sx_gemv: x16_out needs an fp32, non-GLU output with N %% 32 == 0");
  p.x_tiled = a->x_layout;
  SX_CHECK(a->x_layout == 0 || a->x_layout == 1, "sx_gemv: x_layout must be 0 (row-major) or 1 (operand tiles)");
  SX_CHECK(a->x_planes >= 0 && a->x_planes <= 2, "sx_gemv: x_planes=%d", a->x_planes);
  const bool planes2 = a->x_planes == 2;
  p.planes2 = planes2 ? 1 : 0;
  SX_CHECK(!planes2 || a->x_layout == 1, "sx_gemv: x_planes = 2 needs tiled x ([2 planes][row blocks][K/32][16][32])");
  p.out_planes = a->out_planes ? 1 : 0;
  p.x16_gamma = a->x16_gamma;
  SX_CHECK(!p.out_planes || p.y_tiled || a->x16_out, "sx_gemv: out_planes needs a tiled 16-bit output (SX_TILED16 or x16_out)");
  SX_CHECK(!a->x16_gamma || (a->x16_out && (((uintptr_t)a->x16_gamma) & 15) == 0), "sx_gemv: x16_gamma belongs to x16_out (16-B aligned fp32 [N])");
  SX_CHECK(a->w_layout >= 0 && a->w_layout <= 2, "sx_gemv: w_layout must be 0 (row-major), 1 (decode tiles) or 2 (20-row decode tiles)");
  SX_CHECK(a->w_layout != 2 || (!a->glu && a->N % 20 == 0 && a->N % 32 == 0), "sx_gemv: w_layout 2 needs N %% 20 == 0, N %% 32 == 0, no GLU");
  // MI355X (tools/bench_gemv.py, 13B shapes): VALU path 6.5 / 4.7 / 3.0 TB/s at M = 1 / 4 / 8, MFMA path 4.0-4.4 TB/s at any M
  const bool mfma_ok = (a->M >= 2 || planes2) && a->K % 64 == 0 && a->K >= 256 && a->N % 32 == 0;
  SX_CHECK(!a->w_layout || (mfma_ok && a->K % 64 == 0), "sx_gemv: the decode-tile layout needs M >= 2, K %% 64 == 0, K >= 256, N %% 32 == 0");
  // the decode-tile layout only exists for the MFMA kernel: it overrides the VALU test hook
  SX_CHECK(!p.y_tiled || mfma_ok, "sx_gemv: a tiled output needs the MFMA path (M >= 2, K %% 64 == 0, K >= 256, N %% 32 == 0)");
  SX_CHECK(!a->x_layout || mfma_ok, "sx_gemv: tiled x needs the MFMA path (M >= 2, K %% 64 == 0, K >= 256, N %% 32 == 0)");
  if (mfma_ok && (a->w_layout || a->x_layout || p.y_tiled || (g_force_valu_gemv != 1 && (a->M >= 5 || g_force_valu_gemv == 2)))) {
    // MFMA skinny GEMM: R = 2 row groups per wave for GLU (one packed group) or when that still gives >= 256 blocks
    const bool tail20 = a->w_layout == 2;
    const bool r2 = !tail20 && (a->glu || a->N / 32 >= 256);
    // 64-row workgroups (R = 4, one k-step per round): every workgroup re-reads the whole x operand from L2 (64 B x K per 16-row block) —
    // with two planes that traffic is as large as the weight stream itself and its time ADDS to it (tools/bench_skinny_shapes.py:
    // qkv 26 / 32 / 43 us with 1 / 2 / 4 x blocks). Twice the rows per workgroup halve it; only where enough workgroups remain to fill
    // the chip (qkv at 13B: 240, measured faster than 480 32-row ones), and not together with the RMSNorm-fold producer outputs (20-row
    // tiles anyway)
    const bool r4 = r2 && g_skinny_var[3] > 0 && a->N % 64 == 0 && !a->x16_out && (a->N / 64 >= (g_skinny_var[3] == 2 ? 64 : 200));
    const int gx = tail20 ? a->N / 20 : (r4 ? a->N / 64 : (r2 ? a->N / 32 : a->N / 16));
    // split-K over workgroups when the row groups alone do not fill the chip (N = 5120: 320 workgroups on 256 CUs, the CUs
    // with two of them set the time) AND the kernel is long enough to pay for the second pass (publish, count, re-read: ~4 us
    // at the kernel's tail): tools/lab/gemv_lab — down 5120x13824 36.8 -> 33.0 us with S = 4, o 5120x5120 15.4 -> 17.6 (never)
    int S = 1;
    if (a->workspace && !(tail20 && g_skinny_var[2] <= 0)) {      // 256 balanced 20-row groups: no split unless forced (lab)
      if (g_skinny_var[2] > 0) S = g_skinny_var[2];
      else if (g_skinny_var[2] == 0 && a->K >= 8192) while (S < 8 && gx * S < 1024 && (a->K / 64) / (2 * S * 4) >= 4) S *= 2;
      const uint64_t cnt_bytes = 16384;    // fixed, so launches of different N can share one workspace (their partial regions
                                           // may overlap — launches are serial — but never reach the counters)
      if (S > 1 && (gx > 4096 || cnt_bytes + (uint64_t)S * (a->M > 16 ? (planes2 ? 64 : 32) : (planes2 ? 32 : 16)) * a->N * 4 > a->workspace_bytes)) S = 1;   // too small: no split
      p.ws_cnt = (unsigned*)a->workspace;
      p.ws_part = (float*)((char*)a->workspace + cnt_bytes);
    }
    const dim3 grid(gx, S);
#define SX_SK_GO(TT)                                                                                          \
    if (a->M > 16 && planes2 && r4) {                                                                          \
      hipLaunchKernelGGL((gemm_skinny_kernel<TT, 4, 4, 1, false, 4>), grid, dim3(256), 0, ST, p);              \
    } else if (a->M > 16 && planes2) {    /* round 6: 17..32 fp32-grade rows = four operand blocks per weight fragment */      \
      if (tail20) hipLaunchKernelGGL((gemm_skinny_kernel<TT, 2, 4, 1, true, 4>), grid, dim3(256), 0, ST, p);      \
      else if (r2) hipLaunchKernelGGL((gemm_skinny_kernel<TT, 2, 4, 1, false, 4>), grid, dim3(256), 0, ST, p);    \
      else hipLaunchKernelGGL((gemm_skinny_kernel<TT, 1, 4, 2, false, 4>), grid, dim3(256), 0, ST, p);            \
    } else if (r4) {                                                                                           \
      if (a->M > 16 || planes2) hipLaunchKernelGGL((gemm_skinny_kernel<TT, 4, 4, 1, false, 2>), grid, dim3(256), 0, ST, p); \
      else hipLaunchKernelGGL((gemm_skinny_kernel<TT, 4, 4, 2, false, 1>), grid, dim3(256), 0, ST, p);         \
    } else if (a->M > 16 || planes2) {                                                                                \
      if (g_skinny_var[1] == 1) {       /* lab: 4 k-steps per round (256 VGPRs + AGPR copies, one wave per SIMD) */ \
        if (tail20) hipLaunchKernelGGL((gemm_skinny_kernel<TT, 2, 4, 4, true, 2>), grid, dim3(256), 0, ST, p);   \
        else if (r2) hipLaunchKernelGGL((gemm_skinny_kernel<TT, 2, 4, 4, false, 2>), grid, dim3(256), 0, ST, p); \
        else hipLaunchKernelGGL((gemm_skinny_kernel<TT, 1, 4, 4, false, 2>), grid, dim3(256), 0, ST, p);         \
      } else if (tail20 && g_skinny_var[1] == 2) hipLaunchKernelGGL((gemm_skinny_kernel<TT, 2, 4, 4, true, 2>), grid, dim3(256), 0, ST, p); \
      else if (tail20) hipLaunchKernelGGL((gemm_skinny_kernel<TT, 2, 4, 2, true, 2>), grid, dim3(256), 0, ST, p); \
      else if (r2) hipLaunchKernelGGL((gemm_skinny_kernel<TT, 2, 4, 2, false, 2>), grid, dim3(256), 0, ST, p);   \
      else hipLaunchKernelGGL((gemm_skinny_kernel<TT, 1, 4, 4, false, 2>), grid, dim3(256), 0, ST, p);           \
    } else if (tail20) hipLaunchKernelGGL((gemm_skinny_kernel<TT, 2, 4, 4, true>), grid, dim3(256), 0, ST, p); \
    else if (r2) hipLaunchKernelGGL((gemm_skinny_kernel<TT, 2, 4, 4>), grid, dim3(256), 0, ST, p);             \
    else hipLaunchKernelGGL((gemm_skinny_kernel<TT, 1, 4, 4>), grid, dim3(256), 0, ST, p);
    if (a->dtype == SX_BF16) { SX_SK_GO(BF16) } else { SX_SK_GO(F16) }
#undef SX_SK_GO
    SX_HIP_LAUNCH_CHECK();
    return SX_OK;
  }
  SX_CHECK(a->M <= 8, "sx_gemv: M=%d > 8 needs K %% 64 == 0 and N %% 32 == 0 (MFMA path)", a->M);
  SX_CHECK(p.packed == 0, "sx_gemv: the VALU kernel reads row-major weights only");
  SX_CHECK(!a->x16_out && !a->row_ssq_in, "sx_gemv: the RMSNorm fold exists on the MFMA path only");
  const int pairs = (a->N + 1) / 2;
  const dim3 grid((pairs + 3) / 4), block(256);
  const int mr = a->M == 1 ? 1 : (a->M == 2 ? 2 : (a->M <= 4 ? 4 : 8));
#define SX_GEMV_GO(TT)                                                                          \
  switch (mr) {                                                                                 \
    case 1: hipLaunchKernelGGL((gemv_kernel<TT, 1>), grid, block, 0, ST, p); break;             \
    case 2: hipLaunchKernelGGL((gemv_kernel<TT, 2>), grid, block, 0, ST, p); break;             \
    case 4: hipLaunchKernelGGL((gemv_kernel<TT, 4>), grid, block, 0, ST, p); break;             \
    default: hipLaunchKernelGGL((gemv_kernel<TT, 8>), grid, block, 0, ST, p); break;            \
  }
  if (a->dtype == SX_BF16) { SX_GEMV_GO(BF16) } else { SX_GEMV_GO(F16) }
#undef SX_GEMV_GO
  SX_HIP_LAUNCH_CHECK();
  return SX_OK;
}

extern "C" int sx_attn_decode_b(const void* q, const void* kcache, const void* vcache, void* out, float* scratch,
                                const int32_t* ctx_len_dev, int G, int H, int D, int Tmax, int64_t cache_seq_stride,
                                int nsplit, float scale, int dtype, int64_t q_seq_stride, void* stream) {
  SX_CHECK(q && kcache && vcache && out && scratch && ctx_len_dev, "sx_attn_decode: null pointer");
  SX_CHECK(q_seq_stride == 0 || (q_seq_stride >= (int64_t)H * D && q_seq_stride % 8 == 0), "sx_attn_decode: q_seq_stride");
  SX_CHECK(D % 8 == 0 && D <= 128, "sx_attn_decode: head_dim %d", D);
  SX_CHECK(nsplit >= 1 && nsplit <= 64 && G >= 1, "sx_attn_decode: nsplit/G");
  const int tiled = (dtype & SX_TILED16) ? 1 : 0;   // the OUTPUT as operand tiles [H*D/32][16][32] (q and the caches are as always)
  dtype &= 0xff;
  SX_CHECK(!tiled || (G <= 32 && (H * D) % 32 == 0), "sx_attn_decode: SX_TILED16 needs G <= 32 and H*D %% 32 == 0");
  if (dtype == SX_BF16) {
    hipLaunchKernelGGL(attn_decode_kernel<BF16>, dim3(H, nsplit, G), dim3(256), 0, ST, (const unsigned short*)q,
                       (const unsigned short*)kcache, (const unsigned short*)vcache, scratch, ctx_len_dev, D, Tmax,
                       nsplit, scale, (long long)cache_seq_stride, (long long)(q_seq_stride ? q_seq_stride : (int64_t)H * D));
    hipLaunchKernelGGL(attn_decode_combine_kernel<BF16>, dim3(H, G), dim3(128), 0, ST, scratch, (unsigned short*)out, D,
                       nsplit, tiled);
  } else {
    hipLaunchKernelGGL(attn_decode_kernel<F16>, dim3(H, nsplit, G), dim3(256), 0, ST, (const unsigned short*)q,
                       (const unsigned short*)kcache, (const unsigned short*)vcache, scratch, ctx_len_dev, D, Tmax,
                       nsplit, scale, (long long)cache_seq_stride, (long long)(q_seq_stride ? q_seq_stride : (int64_t)H * D));
    hipLaunchKernelGGL(attn_decode_combine_kernel<F16>, dim3(H, G), dim3(128), 0, ST, scratch, (unsigned short*)out, D,
                       nsplit, tiled);
  }
  SX_HIP_LAUNCH_CHECK();
  return SX_OK;
}
extern "C" int sx_attn_decode_fused(const sx_attn_decode_args* a, void* stream) {
  SX_CHECK(a && a->qkv && a->kcache && a->vcache && a->out && a->scratch && a->counters && a->cos_tab && a->sin_tab && a->pos_dev,
           "sx_attn_decode_fused: null pointer");
  const int tiled = (a->dtype & SX_TILED16) ? 1 : 0, dt = a->dtype & 0xff;
  SX_CHECK(dt == SX_F16 || dt == SX_BF16, "sx_attn_decode_fused: dtype");
  SX_CHECK(a->D % 16 == 0 && a->D <= 128, "sx_attn_decode_fused: head_dim %d (must be a multiple of 16, <= 128)", a->D);
  SX_CHECK(a->nsplit >= 1 && a->nsplit <= 64 && a->G >= 1 && a->H >= 1, "sx_attn_decode_fused: nsplit/G/H");
  SX_CHECK(!tiled || (a->G <= 32 && (a->H * a->D) % 32 == 0), "sx_attn_decode_fused: SX_TILED16 needs G <= 32 and H*D %% 32 == 0");
  AttnDecP p;
  p.qkv = (const unsigned short*)a->qkv; p.kc = (unsigned short*)a->kcache; p.vc = (unsigned short*)a->vcache;
  p.out = (unsigned short*)a->out; p.scratch = a->scratch; p.counters = (unsigned*)a->counters;
  p.cos_t = a->cos_tab; p.sin_t = a->sin_tab; p.pos_dev = a->pos_dev;
  p.H = a->H; p.D = a->D; p.Tmax = a->Tmax; p.nsplit = a->nsplit; p.tiled = tiled; p.scale = a->scale;
  p.seq_stride = a->cache_seq_stride;
  const dim3 grid(a->H, a->nsplit, a->G);
  if (dt == SX_BF16) hipLaunchKernelGGL(attn_decode_fused_kernel<BF16>, grid, dim3(256), 0, ST, p);
  else hipLaunchKernelGGL(attn_decode_fused_kernel<F16>, grid, dim3(256), 0, ST, p);
  SX_HIP_LAUNCH_CHECK();
  return SX_OK;
}

extern "C" int sx_attn_decode(const void* q, const void* kcache, const void* vcache, void* out, float* scratch,
                              const int32_t* ctx_len_dev, int H, int D, int Tmax, int nsplit, float scale, int dtype,
                              void* stream) {
  return sx_attn_decode_b(q, kcache, vcache, out, scratch, ctx_len_dev, 1, H, D, Tmax, 0, nsplit, scale, dtype, 0, stream);
}

extern "C" int sx_rope_kv_append_b(void* qkv, void* kcache, void* vcache, const float* cos_tab, const float* sin_tab,
                                   const int32_t* pos0_dev, int G, int T, int H, int D, int Tmax,
                                   int64_t cache_seq_stride, int dtype, void* stream) {
  SX_CHECK(qkv && kcache && vcache && cos_tab && sin_tab && pos0_dev, "sx_rope_kv_append: null pointer");
  SX_CHECK(D % 2 == 0 && G >= 1 && T >= 1, "sx_rope_kv_append: D/G/T");
  const int64_t n = (int64_t)G * T * H * (D / 2);
  if (dtype == SX_BF16)
    hipLaunchKernelGGL(rope_kv_append_kernel<BF16>, gs_grid(n), dim3(256), 0, ST, (unsigned short*)qkv,
                       (unsigned short*)kcache, (unsigned short*)vcache, cos_tab, sin_tab, pos0_dev, T, H, D, Tmax, G,
                       (long long)cache_seq_stride);
  else
    hipLaunchKernelGGL(rope_kv_append_kernel<F16>, gs_grid(n), dim3(256), 0, ST, (unsigned short*)qkv,
                       (unsigned short*)kcache, (unsigned short*)vcache, cos_tab, sin_tab, pos0_dev, T, H, D, Tmax, G,
                       (long long)cache_seq_stride);
  SX_HIP_LAUNCH_CHECK();
  return SX_OK;
}
extern "C" int sx_rope_kv_append(void* qkv, void* kcache, void* vcache, const float* cos_tab, const float* sin_tab,
                                 const int32_t* pos0_dev, int T, int H, int D, int Tmax, int dtype, void* stream) {
  return sx_rope_kv_append_b(qkv, kcache, vcache, cos_tab, sin_tab, pos0_dev, 1, T, H, D, Tmax, 0, dtype, stream);
}

extern "C" int sx_embedding(const int32_t* ids, const void* table, float* out, int T, int dim, int dtype,
                            void* stream) {
  SX_CHECK(ids && table && out, "sx_embedding: null pointer");
  if (dtype == SX_BF16)
    hipLaunchKernelGGL(embedding_kernel<BF16>, gs_grid((int64_t)T * dim), dim3(256), 0, ST, ids,
                       (const unsigned short*)table, out, T, dim);
  else
    hipLaunchKernelGGL(embedding_kernel<F16>, gs_grid((int64_t)T * dim), dim3(256), 0, ST, ids,
                       (const unsigned short*)table, out, T, dim);
  SX_HIP_LAUNCH_CHECK();
  return SX_OK;
}

extern "C" int sx_scatter_rows(const float* src, const int32_t* rows, float* dst, int n, int dim, void* stream) {
  SX_CHECK(src && rows && dst, "sx_scatter_rows: null pointer");
  if (n == 0) return SX_OK;
  hipLaunchKernelGGL(scatter_rows_kernel, gs_grid((int64_t)n * dim), dim3(256), 0, ST, src, rows, dst, n, dim);
  SX_HIP_LAUNCH_CHECK();
  return SX_OK;
}

extern "C" int sx_greedy_next_b(float* logits, int ld_logits, int vocab, const int32_t* img_ids_dev, int n_img,
                                const int32_t* prev_id_dev, int32_t* next_id_dev, int32_t* out_ids, int ld_out,
                                const int32_t* step_dev, int G, void* stream) {
  SX_CHECK(logits && img_ids_dev && prev_id_dev && next_id_dev, "sx_greedy_next: null pointer");
  SX_CHECK(n_img >= 2 && n_img <= 1024 && G >= 1, "sx_greedy_next: n_img=%d G=%d", n_img, G);
  SX_CHECK(!out_ids || step_dev, "sx_greedy_next: out_ids needs step_dev");
  hipLaunchKernelGGL(greedy_next_kernel, dim3(G), dim3(1024), 0, ST, logits, vocab, img_ids_dev, n_img, prev_id_dev,
                     next_id_dev, out_ids, step_dev, ld_logits, ld_out);
  SX_HIP_LAUNCH_CHECK();
  return SX_OK;
}
extern "C" int sx_greedy_next(float* logits, int vocab, const int32_t* img_ids_dev, int n_img,
                              const int32_t* prev_id_dev, int32_t* next_id_dev, int32_t* out_ids,
                              const int32_t* step_dev, void* stream) {
  return sx_greedy_next_b(logits, 0, vocab, img_ids_dev, n_img, prev_id_dev, next_id_dev, out_ids, 0, step_dev, 1, stream);
}

extern "C" int sx_scatter_rows_step(const float* src, const int32_t* step_dev, float* dst, int G, int dim, int seq_rows,
                                    void* stream) {
  SX_CHECK(src && step_dev && dst && G >= 1, "sx_scatter_rows_step: bad args");
  hipLaunchKernelGGL(scatter_rows_step_kernel, gs_grid((int64_t)G * dim), dim3(256), 0, ST, src, step_dev, dst, G, dim,
                     seq_rows);
  SX_HIP_LAUNCH_CHECK();
  return SX_OK;
}
