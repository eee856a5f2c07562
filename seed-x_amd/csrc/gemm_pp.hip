// Ping-pong MFMA GEMM / implicit-GEMM 3x3 convolution tiles for gfx950 (MI355X): 256 x {256, 320} x 64, 8 waves.
//
// Same contract as gemm.hip's lock-step kernel (C = epilogue(A · W^T), include/seedx_hip.h sx_gemm), different schedule.
// The lock-step tile keeps every wave in the same phase: all eight issue their LDS-DMA, then all read fragments, then the
// two waves of each SIMD contend for its matrix pipe — the pipe idles ~50 % of a k-tile (profiles/r2_ab_experiments.md §12).
// Here the block is two groups of four waves (one wave of each group per SIMD) that run ONE BARRIER INTERVAL APART:
// while group 0 is in a matrix segment (16-20 MFMAs, s_setprio 1), group 1 is in a load segment (fragment ds_reads + its
// share of the LDS-DMA for later k-tiles), and vice versa — the CDNA4 "compute / load" role split
// (MI355X_MICROARCH.md §Two waves per SIMD; cdna_hip_programming.md T3+T4+T5).
//
//   wave (g, wc): g = wave >> 2 owns tile rows [128g, 128g+128), wc = wave & 3 owns columns [TN*wc, TN*wc+TN), TN = BN/4.
//   A k-tile (64 deep) is four phases; W fragments of the whole k-tile stay in registers (2*FN*4 VGPRs), A fragments are
//   read per phase (4 m-fragments x one 32-deep k-step = 16 VGPRs):
//     P1: rows 0-63  x k 0-31     P2: rows 0-63  x k 32-63     P3: rows 64-127 x k 0-31     P4: rows 64-127 x k 32-63
//   Each phase = L segment {ds_reads, DMA issue, [counted vmcnt], lgkmcnt(0)} | barrier | M segment {MFMAs} | barrier.
//
// LDS: 2 k-tile buffers x (A 256 rows + W BN rows) x 128 B, rows XOR-swizzled in 16-B chunks exactly like gemm.hip.
// Regions of a buffer are re-filled as soon as BOTH groups have finished reading them, not per whole buffer, which gives
// every DMA a full k-tile (8 barrier intervals) or more in flight:
//     rows A0 (0-63 of each group) and all W rows are last read in L2  → tile u+2's copies are issued in L3 / L4 (X, Y)
//     rows A1 (64-127 of each group) are last read in L4               → tile u+1's copies are issued in L1 / L2 (Za, Zb)
//   and retired by two counted waits per k-tile, both `s_waitcnt vmcnt(4 + NB)` (NB = W slots per wave), placed one full
//   phase before the first read by EITHER group (the other group runs one interval late, so its share needs the margin).
// Every wave issues the same number of DMA instructions per k-tile whatever M, N, K: rows / k-tiles outside the problem
// are fetched through the buffer descriptor's range check (zero fill), so the vmcnt arithmetic has no edge cases.
//
// Epilogue: compile-time specialised on (fp32 | 16-bit output, activation, GLU). The fp32 residual (and nothing else) is
// loaded BEFORE the main loop as the accumulators' initial value, so the epilogue has no loads behind its stores (on gfx9
// loads and stores share vmcnt: a load issued after a store cannot be waited for without the store's acknowledgement —
// the old per-row "load residual, add, store" loop cost 44 us per tile, as long as its main loop).
#include "gemm_common.h"
#include <type_traits>

namespace sxk_gemm {

#define SX_A_CONV3X3_UP 2   // internal: 3x3 conv over a nearest-2x upsampled input (stride 1, pad 1)

#define PP_SYNC()                            \
  do {                                       \
    __builtin_amdgcn_sched_barrier(0);       \
    __builtin_amdgcn_s_barrier();            \
    __builtin_amdgcn_sched_barrier(0);       \
  } while (0)

// LN: 0 = none; 1 = LayerNorm-fold consumer (per-row scale / shift ahead of bias and activation); 2 = producer (fp32 output + its
// 16-bit copy + per-row sums) — see GemmP::ln_*
template <typename TT, int BN, int AMODE, bool OUT32, int ACT, bool GLU, int VAR, int LN = 0>
__global__ __launch_bounds__(512) void gemm_pp_kernel(const GemmP p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef typename TT::vec8 vec8;
  static_assert(LN != 2 || (OUT32 && !GLU && ACT == SX_ACT_NONE), "the LayerNorm producer is the plain fp32-output epilogue");
  constexpr int BM = 256, TN = BN / 4, FN = TN / 16, FM = 8, FH = 4;
  constexpr int NB = BN / 64;                 // W DMA slots (8 rows x 128 B) per wave per k-tile
  constexpr int C1 = NB - 3, C2 = 3;          // W slots issued in L3 (beside the two A0 slots) and in L4
  constexpr int WAITN = 4 + NB;               // DMA instructions allowed to stay in flight at both counted waits
  constexpr int A_BYTES = BM * 128, STAGE = A_BYTES + BN * 128;
  static_assert(!GLU || FN % 2 == 0, "GLU pairs n-fragments");
  static_assert(BN == 256 || BN == 320, "ping-pong tiles are 256x256 and 256x320");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2, wc = wave & 3;
  unsigned long long t_start = 0, t_first = 0, t_main = 0;
  if (p.dbg) t_start = __builtin_amdgcn_s_memtime();

  int tile_m, tile_n;
  if (!tile_coords(p, blockIdx.x, gridDim.x, tile_m, tile_n)) return;  // padding block (exits before any barrier)
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // LayerNorm fold, consumer: the tile's 256 (sum, sum of squares) pairs are requested HERE, ahead of the DMA prologue, and turned
  // into (rstd, -rstd mu) in the 2 KB of LDS behind the operand ring once the prologue is issued — at the epilogue's start the same
  // loads would be an exposed L2 round trip per tile (20 rounds of tiles in the GEGLU projection)
  double ln_s1 = 0.0, ln_s2 = 0.0;
  if constexpr (LN == 1) {
    if (tid < BM) {
      const int m = m0 + tid < p.M ? m0 + tid : p.M - 1;
      ln_s1 = p.ln_in[2 * (size_t)m];
      ln_s2 = p.ln_in[2 * (size_t)m + 1];
    }
  }
  __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)p.a_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (int)p.w_bytes, 0x00020000);

  // ---- per-lane DMA source offsets (fixed over the k loop) ---------------------------------------------------------
  // A slots of this wave: s = 0,1 → rows 128g + ((2*wave + s) & 7) * 8 of A0; s = 2,3 → the same + 64 (A1)
  const int rl = lane >> 3;                         // row within the 8-row slot
  const unsigned gchunk = ((lane & 7) ^ rl) << 4;   // swizzled 16-B chunk this lane fetches
  // LINEAR: a_off = byte offset of the lane's row (+ swizzled chunk); rows >= M carry 0xC0000000 (beyond any 2-GiB descriptor,
  // and still beyond it after the k offset is added).
  // CONV: a_off = byte offset of filter tap (0, 0) of the lane's output pixel (may be "negative": only used for valid taps),
  //       a_msk = bit t set iff tap t = 3*dy + dx reads inside the image (and the row is < M); with nearest-2x upsampling
  //       bits 9 / 10 hold the parity of the output row / column: the source pixel of tap (dy, dx) is then
  //       ((oy>>1) + ((py + dy - 1) >> 1), (ox>>1) + ((px + dx - 1) >> 1)) — no per-tap divisions or branches in the loop
  unsigned a_off[4], a_msk[4];
  int a_lrow[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    a_lrow[s] = g * 128 + (s >> 1) * 64 + ((2 * wave + (s & 1)) & 7) * 8;
    const int row = m0 + a_lrow[s] + rl;
    if (AMODE == SX_A_LINEAR) {
      a_off[s] = (row < p.M) ? (unsigned)row * (unsigned)p.K * 2u + gchunk : 0xC0000000u;
      a_msk[s] = 0;
    } else {
      const int hw = p.Hout * p.Wout;
      const int b = row / hw, rem = row - b * hw;
      const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
      const int Hv = p.upsample ? 2 * p.Hin : p.Hin, Wv = p.upsample ? 2 * p.Win : p.Win;
      const int vy = oy * p.stride - p.pad, vx = ox * p.stride - p.pad;   // tap (0,0) in (virtual) input coordinates
      unsigned msk = 0;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int iy = vy + t / 3, ix = vx + t % 3;
        if (row < p.M && iy >= 0 && iy < Hv && ix >= 0 && ix < Wv) msk |= 1u << t;
      }
      int sy = vy, sx = vx;
      if (AMODE == SX_A_CONV3X3_UP) {       // stride 1, pad 1: vy = oy - 1
        sy = oy >> 1; sx = ox >> 1;
        msk |= (unsigned)(oy & 1) << 9 | (unsigned)(ox & 1) << 10;
      }
      a_off[s] = (unsigned)(((b * p.Hin + sy) * p.Win + sx) * p.Cin) * 2u + gchunk;
      a_msk[s] = msk;
    }
  }
  unsigned w_off[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int row = n0 + (wave * NB + i) * 8 + rl;
    w_off[i] = (row < p.N) ? (unsigned)row * (unsigned)p.Kw * 2u + gchunk : 0xC0000000u;
  }
  const int cpt = (AMODE != SX_A_LINEAR) ? p.Cin / 64 : 1;  // k-tiles per filter tap
  const int nkt = p.K / 64;
  const int nkw = p.Kw / 64;   // k-tiles of W: = nkt, or nkt / 2 when A is [hi | lo] planes (sx_gemm_args.a_planes = 2: W walked twice)

  // filter-tap walkers of the two A streams (conv only): kx = state of the k-tile whose A0 rows are issued next (u + 2),
  // kz = state of the k-tile whose A1 rows are issued next (u + 1). Advanced incrementally: no division in the loop.
  struct Tap { int cc, dy, dx; };
  auto tap_next = [&](Tap& t) {
    if (++t.cc == cpt) { t.cc = 0; if (++t.dx == 3) { t.dx = 0; ++t.dy; } }
  };

  // k-tile offset shared by every DMA of a k-tile: kt * 128 B, or 2 GiB for k-tiles past the end (zero fill, never read)
  auto koff_of = [&](int kt) -> unsigned { return (kt < nkt) ? (unsigned)kt * 128u : 0x80000000u; };
  auto koff_w = [&](int kt) -> unsigned { return (kt < nkt) ? (unsigned)(kt >= nkw ? kt - nkw : kt) * 128u : 0x80000000u; };
  auto dma_a = [&](int s, int buf, int kt, const Tap& t) {
    unsigned voff;
    if (AMODE == SX_A_LINEAR) {
      voff = a_off[s] + koff_of(kt);
    } else {
      const int tap = t.dy * 3 + t.dx;                                  // wave-uniform
      const bool ok = ((a_msk[s] >> tap) & 1u) != 0 && kt < nkt;
      unsigned o;
      if (AMODE == SX_A_CONV3X3_UP) {
        const int ry = ((int)((a_msk[s] >> 9) & 1u) + t.dy - 1) >> 1, rx = ((int)((a_msk[s] >> 10) & 1u) + t.dx - 1) >> 1;
        o = a_off[s] + (unsigned)((ry * p.Win + rx) * p.Cin) * 2u + (unsigned)t.cc * 128u;
      } else {
        o = a_off[s] + (unsigned)(((t.dy * p.Win + t.dx) * p.Cin + t.cc * 64) * 2);   // uniform delta
      }
      voff = ok ? o : 0x80000000u;
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, SX_LDS_PTR(smem + buf * STAGE + a_lrow[s] * 128), 16, voff, 0, 0, 0);
  };
  auto dma_w = [&](int i, int buf, int kt) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, SX_LDS_PTR(smem + buf * STAGE + A_BYTES + (wave * NB + i) * 1024), 16,
                                             w_off[i] + koff_w(kt), 0, 0, 0);
  };
  // X(kt): A0 slots + first C1 W slots; Y(kt): remaining W slots; Za / Zb(kt): the two A1 slots
  auto issue_x = [&](int buf, int kt, const Tap& t) {
    dma_a(0, buf, kt, t);
    dma_a(1, buf, kt, t);
#pragma unroll
    for (int i = 0; i < C1; ++i) dma_w(i, buf, kt);
  };
  auto issue_y = [&](int buf, int kt) {
#pragma unroll
    for (int i = C1; i < NB; ++i) dma_w(i, buf, kt);
  };

  // ---- accumulators ------------------------------------------------------------------------------------------------
  // lane holds C[m][n .. n+3], m = ..+(lane & 15), n = ..+(lane >> 4)*4   (operands swapped in the MFMA: D = Wfrag · Afrag^T)
  f32x4_t acc[FN][FM];
  const int lq = (lane >> 4) * 4;
  if (p.res_init) {
    // fp32 residual as the initial value: 16 rows x 64 B per load instruction; issued ahead of the DMA prologue
#pragma unroll
    for (int j = 0; j < FM; ++j) {
      const int m = m0 + g * 128 + j * 16 + (lane & 15);
      const int mc = m < p.M ? m : p.M - 1;
      const float* rr = p.residual + (size_t)(p.res_mod ? (mc % p.res_mod) : mc) * p.ldr;
#pragma unroll
      for (int i = 0; i < FN; ++i) {
        int c = n0 + wc * TN + i * 16 + lq;
        if (c + 4 > p.ldr) c = 0;
        acc[i][j] = *(const f32x4_t*)(rr + c);
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
      for (int j = 0; j < FM; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }

  // fragment read offsets: row = base + (lane & 15), logical chunk = ks*4 + (lane >> 4), key = lane & 7
  const unsigned frag_row = (unsigned)(lane & 15) * 128u;
  unsigned frag_sw[2];
  frag_sw[0] = (unsigned)(((lane >> 4)) ^ (lane & 7)) << 4;
  frag_sw[1] = (unsigned)((4 + (lane >> 4)) ^ (lane & 7)) << 4;
  const unsigned a_frag = (unsigned)(g * 128) * 128u + frag_row;                 // + h*8192 + j*2048 + frag_sw[ks]
  const unsigned w_frag = (unsigned)A_BYTES + (unsigned)(wc * TN) * 128u + frag_row;  // + i*2048 + frag_sw[ks]

  // ---- prologue: X(0) Y(0) Za(0) Zb(0) X(1) Y(1) -------------------------------------------------------------------
  Tap kx = {0, 0, 0}, kz = {0, 0, 0};
  issue_x(0, 0, kx);
  issue_y(0, 0);
  dma_a(2, 0, 0, kz);
  dma_a(3, 0, 0, kz);
  tap_next(kz);          // kz → k-tile 1
  tap_next(kx);          // kx → k-tile 1
  issue_x(1, 1, kx);
  issue_y(1, 1);
  tap_next(kx);          // kx → k-tile 2
  float* ln_lds = (float*)(smem + 2 * STAGE);
  if constexpr (LN == 1) {
    if (tid < BM) {
      const double mu = ln_s1 * (double)p.ln_inv_dim;
      const double var = ln_s2 * (double)p.ln_inv_dim - mu * mu;
      const float rstd = __builtin_amdgcn_rsqf((float)(var > 0.0 ? var : 0.0) + p.ln_eps);
      *(f32x2_t*)(ln_lds + 2 * tid) = (f32x2_t){rstd, -rstd * (float)mu};      // read in the epilogue, many barriers later
    }
  }
  wait_vmcnt<WAITN>();   // X(0), Y(0) landed (this wave's share)
  PP_SYNC();
  if (p.dbg) t_first = __builtin_amdgcn_s_memtime();
  if (g == 1 && VAR != 3) PP_SYNC();   // group 1 runs one barrier interval behind group 0 (VAR 3: lock-step, A/B only)

  vec8 af[FH], wf0[FN], wf1[FN];

  auto mma = [&](const vec8* wf, int h) {
    if (VAR != 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
      for (int j = 0; j < FH; ++j) acc[i][h * FH + j] = TT::mfma16(wf[i], af[j], acc[i][h * FH + j]);
    if (VAR != 1) __builtin_amdgcn_s_setprio(0);
  };
  auto lgkm0 = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };

  // one k-tile (u) out of ring buffer BUF (compile-time so every LDS address is base + immediate)
  auto ktile = [&](auto BUFC, int u) {
    constexpr int BUF = decltype(BUFC)::value;
    const unsigned char* sbuf = smem + BUF * STAGE;
    // ---- P1: rows 0-63 x k 0-31 ; Za(u+1) ----
#pragma unroll
    for (int j = 0; j < FH; ++j) af[j] = *(const vec8*)(sbuf + a_frag + j * 2048 + frag_sw[0]);
#pragma unroll
    for (int i = 0; i < FN; ++i) wf0[i] = *(const vec8*)(sbuf + w_frag + i * 2048 + frag_sw[0]);
    dma_a(2, BUF ^ 1, u + 1, kz);
    if (VAR != 2) lgkm0();
    PP_SYNC();
    mma(wf0, 0);
    PP_SYNC();
    // ---- P2: rows 0-63 x k 32-63 ; Zb(u+1) ; A1(u) landed ----
#pragma unroll
    for (int j = 0; j < FH; ++j) af[j] = *(const vec8*)(sbuf + a_frag + j * 2048 + frag_sw[1]);
#pragma unroll
    for (int i = 0; i < FN; ++i) wf1[i] = *(const vec8*)(sbuf + w_frag + i * 2048 + frag_sw[1]);
    dma_a(3, BUF ^ 1, u + 1, kz);
    tap_next(kz);
    wait_vmcnt<WAITN>();
    if (VAR != 2) lgkm0();
    PP_SYNC();
    mma(wf1, 0);
    PP_SYNC();
    // ---- P3: rows 64-127 x k 0-31 ; X(u+2) ----
#pragma unroll
    for (int j = 0; j < FH; ++j) af[j] = *(const vec8*)(sbuf + a_frag + 8192 + j * 2048 + frag_sw[0]);
    issue_x(BUF, u + 2, kx);
    if (VAR != 2) lgkm0();
    PP_SYNC();
    mma(wf0, 1);
    PP_SYNC();
    // ---- P4: rows 64-127 x k 32-63 ; Y(u+2) ; A0(u+1), W(u+1) landed ----
#pragma unroll
    for (int j = 0; j < FH; ++j) af[j] = *(const vec8*)(sbuf + a_frag + 8192 + j * 2048 + frag_sw[1]);
    issue_y(BUF, u + 2);
    tap_next(kx);
    wait_vmcnt<WAITN>();
    if (VAR != 2) lgkm0();
    PP_SYNC();
    mma(wf1, 1);
    PP_SYNC();
  };

  for (int kt = 0; kt < nkt; kt += 2) {
    ktile(std::integral_constant<int, 0>{}, kt);
    if (kt + 1 < nkt) ktile(std::integral_constant<int, 1>{}, kt + 1);
  }
  if (g == 0 && VAR != 3) PP_SYNC();   // pairs with group 1's last barrier
  wait_vmcnt<0>();         // the zero-fill DMAs of the k-tiles past the end must not outlive the block's LDS allocation

  if (p.dbg) t_main = __builtin_amdgcn_s_memtime();
  // ---- fused epilogue (no loads behind stores) ---------------------------------------------------------------------
  int ncol[FN], nout[FN];
  bool nok[FN];
  f32x4_t bv[FN];
#pragma unroll
  for (int i = 0; i < FN; ++i) {
    const int nb = n0 + wc * TN + i * 16;
    ncol[i] = nb + lq;
    nout[i] = GLU ? (nb >> 1) + lq : ncol[i];
    nok[i] = ncol[i] < p.N && nout[i] < p.n_valid && !(GLU && (i & 1));
    if (ncol[i] >= p.N) ncol[i] = p.N - 4;
    if (nout[i] + 4 > p.ldc) nout[i] = 0;  // clamped lanes never store
    bv[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }
  if (p.bias) {
#pragma unroll
    for (int i = 0; i < FN; ++i) bv[i] = *(const f32x4_t*)(p.bias + ncol[i]);
  }
  // per-sample bias rows (time-embedding add of the resnet convs): a tile inside one sample adds it to the bias vector once
  const bool b2_uniform = p.bias2d && (m0 / p.bias2d_rows) == ((m0 + BM - 1 < p.M ? m0 + BM - 1 : p.M - 1) / p.bias2d_rows);
  if (b2_uniform) {
    const float* b2 = p.bias2d + (size_t)(m0 / p.bias2d_rows) * p.ldb2;
#pragma unroll
    for (int i = 0; i < FN; ++i) bv[i] += *(const f32x4_t*)(b2 + ncol[i]);
  }
  const bool b2_rows = p.bias2d && !b2_uniform;
  const bool res_late = p.residual && !p.res_init;   // activation + residual (not on the hot path): loads inside the row loop
  const bool wide = !OUT32 && !GLU && (p.ldc & 7) == 0 && (((size_t)p.C) & 15) == 0;
  bool pair_ok[(FN + 1) / 2];
#pragma unroll
  for (int i = 0; i < (FN + 1) / 2; ++i) {
    const int lim = p.n_valid < p.N ? p.n_valid : p.N;
    pair_ok[i] = (2 * i + 1 < FN) && (n0 + wc * TN + i * 32 + 32 <= lim);  // wave-uniform: the whole 32-col pair is stored
  }
  // GLU outputs of this wave: columns [(n0 + wc TN) / 2, + 32) of the [M][N / 2] output — wide stores when all 32 exist and are aligned
  const bool glu_wide = GLU && !OUT32 && !(p.tune & 1) && (p.ldc & 7) == 0 && (((size_t)p.C) & 15) == 0 &&
                        ((n0 + wc * TN) >> 1) + 32 <= (p.n_valid < (p.N >> 1) ? p.n_valid : (p.N >> 1));
  auto act1 = [&](f32x4_t x) -> f32x4_t {
    if (ACT == SX_ACT_GELU) {
      const f32x2_t g0 = gelu_erf2((f32x2_t){x[0], x[1]}), g1 = gelu_erf2((f32x2_t){x[2], x[3]});
      return (f32x4_t){g0[0], g0[1], g1[0], g1[1]};
    } else if (ACT == SX_ACT_SILU) {
      return (f32x4_t){silu_f(x[0]), silu_f(x[1]), silu_f(x[2]), silu_f(x[3])};
    }
    return x;
  };
  auto pack4 = [&](f32x4_t x) -> u32x2_t {
    u32x2_t o;
    o[0] = pack2<TT>(x[0], x[1]);
    o[1] = pack2<TT>(x[2], x[3]);
    return o;
  };
  // LayerNorm fold, consumer: the row loop reads (rstd, -rstd mu) of its rows from the LDS table written during the prologue
  f32x4_t lcs[LN == 1 ? FN : 1];
  if constexpr (LN == 1) {
#pragma unroll
    for (int i = 0; i < FN; ++i) lcs[i] = *(const f32x4_t*)(p.ln_cs + ncol[i]);
  }
  float lsum[LN == 2 ? FM : 1], lsq[LN == 2 ? FM : 1];
  // fused GroupNorm statistics (fp32-output kernels only): per lane the sums of its two column PAIRS of every n-fragment over its 8
  // rows — a pair never straddles a group (channels per group are even), a quad can (C = 320: 10 channels per group)
  const bool gn = OUT32 && p.gn_stats != nullptr;
  float gsl[FN], gql[FN], gsh[FN], gqh[FN];
#pragma unroll
  for (int i = 0; i < FN; ++i) gsl[i] = gql[i] = gsh[i] = gqh[i] = 0.f;
#pragma unroll
  for (int j = 0; j < FM; ++j) {
    const int m = m0 + g * 128 + j * 16 + (lane & 15);
    const bool mok = m < p.M;
    const int mc = mok ? m : p.M - 1;
    f32x4_t v[FN];
    if constexpr (LN == 1) {
      const f32x2_t ab = *(const f32x2_t*)(ln_lds + 2 * (g * 128 + j * 16 + (lane & 15)));
#pragma unroll
      for (int i = 0; i < FN; ++i) v[i] = acc[i][j] * ab[0] + (lcs[i] * ab[1] + bv[i]);
    } else {
#pragma unroll
      for (int i = 0; i < FN; ++i) v[i] = acc[i][j] + bv[i];
    }
    if (b2_rows) {
      const float* b2 = p.bias2d + (size_t)(mc / p.bias2d_rows) * p.ldb2;
#pragma unroll
      for (int i = 0; i < FN; ++i) v[i] += *(const f32x4_t*)(b2 + ncol[i]);
    }
    if (GLU) {
#pragma unroll
      for (int i = 0; i + 1 < FN; i += 2) {
        const f32x4_t gt = act1(v[i + 1]);
        v[i] = v[i] * gt;
      }
    } else if (ACT != SX_ACT_NONE) {
#pragma unroll
      for (int i = 0; i < FN; ++i) v[i] = act1(v[i]);
    }
    if (res_late) {
      const float* rr = p.residual + (size_t)(p.res_mod ? (mc % p.res_mod) : mc) * p.ldr;
#pragma unroll
      for (int i = 0; i < FN; ++i) {
        if (GLU && (i & 1)) continue;
        const int c = (nout[i] + 4 <= p.ldr) ? nout[i] : 0;
        v[i] += *(const f32x4_t*)(rr + c);
      }
    }
    if (OUT32) {
#pragma unroll
      for (int i = 0; i < FN; ++i)
        if (mok && nok[i]) *(f32x4_t*)((float*)p.C + (size_t)m * p.ldc + nout[i]) = v[i];
      if constexpr (LN == 2) {
        // the row's 16-bit copy (the operand of the GEMM behind the folded LayerNorm) and this lane's share of its two sums
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < FN; ++i) {
          if (mok && nok[i]) {
            s1 += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
            s2 += (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]);
          }
        }
        lsum[j] = s1;
        lsq[j] = s2;
        const bool xwide = (p.ln_ldx & 7) == 0 && (((size_t)p.ln_x16) & 15) == 0;
#pragma unroll
        for (int i = 0; i < FN; ++i) {
          if ((i & 1) == 0 && i + 1 < FN && xwide && pair_ok[i >> 1]) {      // 16-B stores, as in the 16-bit epilogue below
            const u32x2_t o0 = pack4(v[i]), o1 = pack4(v[i + 1]);
            const auto w0 = __builtin_amdgcn_permlane16_swap(o0[0], o1[0], false, false);
            const auto w1 = __builtin_amdgcn_permlane16_swap(o0[1], o1[1], false, false);
            if (mok) {
              const u32x4_t w4 = {w0[0], w1[0], w0[1], w1[1]};
              const int q = lane >> 4;
              const int col = n0 + wc * TN + i * 16 + (q & 1) * 16 + (q >> 1) * 8;
              *(u32x4_t*)((unsigned short*)p.ln_x16 + (size_t)m * p.ln_ldx + col) = w4;
            }
            continue;
          }
          if ((i & 1) && xwide && pair_ok[i >> 1]) continue;                   // stored with its even partner
          if (mok && nok[i]) *(u32x2_t*)((unsigned short*)p.ln_x16 + (size_t)m * p.ln_ldx + nout[i]) = pack4(v[i]);
        }
      }
      if (gn && mok) {
#pragma unroll
        for (int i = 0; i < FN; ++i) {
          gsl[i] += v[i][0] + v[i][1];
          gql[i] += v[i][0] * v[i][0] + v[i][1] * v[i][1];
          gsh[i] += v[i][2] + v[i][3];
          gqh[i] += v[i][2] * v[i][2] + v[i][3] * v[i][3];
        }
      }
    } else {
      if constexpr (GLU && FN == 4) {
        // GLU: the wave's two results (fragment pairs (0, 1) and (2, 3)) are two ADJACENT 16-column output blocks, 4 columns per
        // lane each → the same v_permlane16_swap trade as the plain 16-bit epilogue gives every lane 8 consecutive columns: one
        // 16-byte store per row instead of two 8-byte stores (the store tail of a tile is issue-bound, not byte-bound)
        if (glu_wide) {
          const u32x2_t o0 = pack4(v[0]), o1 = pack4(v[2]);
          const auto s0 = __builtin_amdgcn_permlane16_swap(o0[0], o1[0], false, false);
          const auto s1 = __builtin_amdgcn_permlane16_swap(o0[1], o1[1], false, false);
          if (mok) {
            const u32x4_t w4 = {s0[0], s1[0], s0[1], s1[1]};
            const int q = lane >> 4;
            const int col = ((n0 + wc * TN) >> 1) + (q & 1) * 16 + (q >> 1) * 8;
            *(u32x4_t*)((unsigned short*)p.C + (size_t)m * p.ldc + col) = w4;
          }
          continue;
        }
      }
#pragma unroll
      for (int i = 0; i < FN; ++i) {
        if (GLU && (i & 1)) continue;
        if (!GLU && wide && (i & 1) == 0 && i + 1 < FN && pair_ok[i >> 1]) {
          // two neighbouring n-fragments hold cols [nb, nb+16) and [nb+16, nb+32) as 4 per lane. v_permlane16_swap trades the
          // odd 16-lane rows of fragment i with the even rows of fragment i+1, after which a lane owns 8 CONSECUTIVE columns
          // → one 16-B store instead of two 8-B stores
          const u32x2_t o0 = pack4(v[i]), o1 = pack4(v[i + 1]);
          const auto s0 = __builtin_amdgcn_permlane16_swap(o0[0], o1[0], false, false);
          const auto s1 = __builtin_amdgcn_permlane16_swap(o0[1], o1[1], false, false);
          if (mok) {
            const u32x4_t w4 = {s0[0], s1[0], s0[1], s1[1]};
            const int q = lane >> 4;
            const int col = n0 + wc * TN + i * 16 + (q & 1) * 16 + (q >> 1) * 8;
            *(u32x4_t*)((unsigned short*)p.C + (size_t)m * p.ldc + col) = w4;
          }
          continue;
        }
        if (!GLU && wide && (i & 1) && pair_ok[i >> 1]) continue;  // stored with its even partner
        if (mok && nok[i]) *(u32x2_t*)((unsigned short*)p.C + (size_t)m * p.ldc + nout[i]) = pack4(v[i]);
      }
    }
  }
  if constexpr (LN == 2) {
    // per-row sums: the four 16-lane rows of a wave hold different column quads of the same 16 tile rows → two xor shuffles, one
    // LDS slot per (column wave, row), the four column waves added in a fixed order, one fp64 atomic pair per row and tile
    float* racc = (float*)smem;                    // [4 wc][256 rows][2]
    __syncthreads();
#pragma unroll
    for (int j = 0; j < FM; ++j) {
      float s1 = lsum[j], s2 = lsq[j];
      s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
      s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
      if (lane < 16) *(f32x2_t*)(racc + ((size_t)(wc * BM + g * 128 + j * 16 + lane)) * 2) = (f32x2_t){s1, s2};
    }
    __syncthreads();
    if (tid < BM && m0 + tid < p.M) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { s1 += racc[(w * BM + tid) * 2]; s2 += racc[(w * BM + tid) * 2 + 1]; }
      atomicAdd(&p.ln_out[2 * (size_t)(m0 + tid)], (double)s1);
      atomicAdd(&p.ln_out[2 * (size_t)(m0 + tid) + 1], (double)s2);
    }
  }
  if (OUT32 && gn) {
    // 16 lanes (lane & 15 = the rows of a fragment) → one; then one LDS accumulator pair per group the tile touches (the
    // operand ring is dead: every wave has drained its DMAs and passed its last fragment read), then one fp64 atomic per value.
    // A 256-row tile lies inside one sample (host: gn_rows % 256 == 0).
    // sum over the 16 lanes of a DPP row with four DPP moves (no LDS traffic): quad_perm [1,0,3,2], quad_perm [2,3,0,1], then
    // row_half_mirror and row_mirror — after the two quad steps the four lanes of a quad agree, so the mirrors act as xor 4 / xor 8
    auto row16_sum = [](float x) -> float {
      x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
      x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));
      x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));
      x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true));
      return x;
    };
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      gsl[i] = row16_sum(gsl[i]);
      gql[i] = row16_sum(gql[i]);
      gsh[i] = row16_sum(gsh[i]);
      gqh[i] = row16_sum(gqh[i]);
    }
    // FIXED-ORDER reduction inside the tile (round 6; LDS float atomics met in arrival order before: the tile's partial sums — and with
    // them mean / rstd, and through the 16-bit roundings behind them the whole forward — differed from run to run): every contributing
    // lane writes its column pair's sums to its own slot [row group][pair], one thread per (group, moment) then adds the group's slots
    // in index order. The tiles still meet in global memory by fp64 atomics: a sum of fp32-valued terms is exact in fp64 (as long as
    // the terms span < 2^29), hence order-independent.
    float* gslot = (float*)smem;                                     // [2 row groups][BN / 2 pairs][2]
    const int gbase = n0 / p.gn_cpg;
    const int nlast = (n0 + BN < p.N ? n0 + BN : p.N) - 1;
    const int ngr = nlast / p.gn_cpg - gbase + 1;                    // groups this tile contributes to (<= BN / 2 + 1)
    __syncthreads();
    if ((lane & 15) == 0) {
#pragma unroll
      for (int i = 0; i < FN; ++i) {
        const int cp = (wc * TN + i * 16 + lq) >> 1;
        *(f32x4_t*)(gslot + ((size_t)g * (BN / 2) + cp) * 2) = (f32x4_t){gsl[i], gql[i], gsh[i], gqh[i]};   // pairs cp, cp + 1
      }
    }
    __syncthreads();
    const int smp = m0 / p.gn_rows;
    for (int t = tid; t < 2 * ngr; t += 512) {
      const int G = gbase + (t >> 1), which = t & 1;
      int c_lo = G * p.gn_cpg - n0, c_hi = (G + 1) * p.gn_cpg - n0;              // the group's columns inside this tile
      c_lo = c_lo < 0 ? 0 : c_lo;
      const int c_end = (p.N - n0 < BN ? p.N - n0 : BN);
      c_hi = c_hi > c_end ? c_end : c_hi;
      float acc = 0.f;
      for (int cp = c_lo >> 1; cp < (c_hi >> 1); ++cp) acc += gslot[(size_t)cp * 2 + which];
      for (int cp = c_lo >> 1; cp < (c_hi >> 1); ++cp) acc += gslot[((size_t)(BN / 2) + cp) * 2 + which];
      atomicAdd(&p.gn_stats[((size_t)smp * p.gn_groups + G) * 2 + which], (double)acc);
    }
  }
  if (p.dbg) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the tile's stores have been issued and accepted
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    if (tid == 0) {
      unsigned long long* d = p.dbg + (size_t)blockIdx.x * 4;
      d[0] = t_start; d[1] = t_first; d[2] = t_main; d[3] = t_end;
    }
  }
#endif
}

int g_pp_variant = 0;
int g_tune = 0;

template <typename TT, int BN, int AMODE, bool OUT32, int ACT, bool GLU, int VAR, int LN = 0>
static int launch_one(const GemmP& p0, hipStream_t st) {
  GemmP p = p0;
  p.dbg = g_dbg;
  p.tune = g_tune;
  const int grid = plan_grid(p, 256, BN, g_xcd_2d, g_gm);
  constexpr size_t lds = 2 * (size_t)(256 + BN) * 128 + (LN == 1 ? 2048 : 0);      // + the consumer's (rstd, -rstd mu) table
  auto k = gemm_pp_kernel<TT, BN, AMODE, OUT32, ACT, GLU, VAR, LN>;
  static hipError_t attr[16];
  static bool done[16];
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev &= 15;
  if (!done[dev]) {
    attr[dev] = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    done[dev] = true;
  }
  SX_CHECK(attr[dev] == hipSuccess, "sx_gemm: cannot reserve %zu B of LDS for the 256x%d ping-pong tile: %s", lds, BN,
           hipGetErrorString(attr[dev]));
  hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, st, p);
  SX_HIP_LAUNCH_CHECK();
  return SX_OK;
}

// epilogue combinations on the hot path (everything else runs on the lock-step kernel): 16-bit out x {none, GELU,
// GELU-GLU, SiLU-GLU}, fp32 out x none. GLU needs an even fragment count (256-wide tile only); convs have no activation.
static int epi_code(const GemmP& p, int dtype) {
  const bool out32 = p.out_dtype == SX_F32;
  if (!out32 && p.out_dtype != dtype) return -1;
  if (p.glu) {
    if (out32) return -1;
    if (p.act == SX_ACT_GELU) return 2;
    if (p.act == SX_ACT_SILU) return 3;
    return -1;
  }
  if (p.act == SX_ACT_NONE) return out32 ? 4 : 0;
  if (p.act == SX_ACT_GELU && !out32) return 1;
  return -1;
}

bool pp_supported(const GemmP& p, int dtype, int bn, int a_mode) {
  const int e = epi_code(p, dtype);
  if (e < 0) return false;
  // LayerNorm fold: consumers = the 16-bit linear epilogues behind a LayerNorm (plain, GELU-GLU), producer = fp32 linear output
  if (p.ln_in && !(a_mode == SX_A_LINEAR && (e == 0 || e == 2) && !p.ln_out)) return false;
  if (p.ln_out && !(a_mode == SX_A_LINEAR && e == 4 && !p.gn_stats)) return false;
  if (bn == 320 && (e == 2 || e == 3)) return false;
  if (a_mode == SX_A_CONV3X3 && !(e == 0 || e == 4)) return false;
  if (a_mode == SX_A_CONV3X3 && p.upsample && !(p.stride == 1 && p.pad == 1)) return false;
  return bn == 256 || bn == 320;
}

template <typename TT>
static int launch_t(const GemmP& p, int dtype, int bn, int a_mode, hipStream_t st) {
  const int e = epi_code(p, dtype);
#define PP_CASE(BNV, AM, O32, ACTV, GLUV) return launch_one<TT, BNV, AM, O32, ACTV, GLUV, 0>(p, st)
  if (p.ln_in) {
    if (e == 2) return launch_one<TT, 256, SX_A_LINEAR, false, SX_ACT_GELU, true, 0, 1>(p, st);
    if (bn == 256) return launch_one<TT, 256, SX_A_LINEAR, false, SX_ACT_NONE, false, 0, 1>(p, st);
    return launch_one<TT, 320, SX_A_LINEAR, false, SX_ACT_NONE, false, 0, 1>(p, st);
  }
  if (p.ln_out) {
    if (bn == 256) return launch_one<TT, 256, SX_A_LINEAR, true, SX_ACT_NONE, false, 0, 2>(p, st);
    return launch_one<TT, 320, SX_A_LINEAR, true, SX_ACT_NONE, false, 0, 2>(p, st);
  }
  if (a_mode == SX_A_LINEAR) {
    if (bn == 256) {
      if (std::is_same<TT, BF16>::value && e == 0 && g_pp_variant == 1) return launch_one<BF16, 256, SX_A_LINEAR, false, SX_ACT_NONE, false, 1>(p, st);
      if (std::is_same<TT, BF16>::value && e == 0 && g_pp_variant == 2) return launch_one<BF16, 256, SX_A_LINEAR, false, SX_ACT_NONE, false, 2>(p, st);
      if (std::is_same<TT, BF16>::value && e == 0 && g_pp_variant == 3) return launch_one<BF16, 256, SX_A_LINEAR, false, SX_ACT_NONE, false, 3>(p, st);
      switch (e) {
        case 0: PP_CASE(256, SX_A_LINEAR, false, SX_ACT_NONE, false);
        case 1: PP_CASE(256, SX_A_LINEAR, false, SX_ACT_GELU, false);
        case 2: PP_CASE(256, SX_A_LINEAR, false, SX_ACT_GELU, true);
        case 3: PP_CASE(256, SX_A_LINEAR, false, SX_ACT_SILU, true);
        case 4: PP_CASE(256, SX_A_LINEAR, true, SX_ACT_NONE, false);
      }
    } else {
      switch (e) {
        case 0: PP_CASE(320, SX_A_LINEAR, false, SX_ACT_NONE, false);
        case 1: PP_CASE(320, SX_A_LINEAR, false, SX_ACT_GELU, false);
        case 4: PP_CASE(320, SX_A_LINEAR, true, SX_ACT_NONE, false);
      }
    }
  } else if (!p.upsample) {
    if (bn == 256) {
      switch (e) {
        case 0: PP_CASE(256, SX_A_CONV3X3, false, SX_ACT_NONE, false);
        case 4: PP_CASE(256, SX_A_CONV3X3, true, SX_ACT_NONE, false);
      }
    } else {
      switch (e) {
        case 0: PP_CASE(320, SX_A_CONV3X3, false, SX_ACT_NONE, false);
        case 4: PP_CASE(320, SX_A_CONV3X3, true, SX_ACT_NONE, false);
      }
    }
  } else {
    if (bn == 256) {
      switch (e) {
        case 0: PP_CASE(256, SX_A_CONV3X3_UP, false, SX_ACT_NONE, false);
        case 4: PP_CASE(256, SX_A_CONV3X3_UP, true, SX_ACT_NONE, false);
      }
    } else {
      switch (e) {
        case 0: PP_CASE(320, SX_A_CONV3X3_UP, false, SX_ACT_NONE, false);
        case 4: PP_CASE(320, SX_A_CONV3X3_UP, true, SX_ACT_NONE, false);
      }
    }
  }
#undef PP_CASE
  return -1;
}

int launch_pp(const GemmP& p, int dtype, int bn, int a_mode, hipStream_t st) {
  if (!pp_supported(p, dtype, bn, a_mode)) return -1;
  return dtype == SX_BF16 ? launch_t<BF16>(p, dtype, bn, a_mode, st) : launch_t<F16>(p, dtype, bn, a_mode, st);
}

}  // namespace sxk_gemm
