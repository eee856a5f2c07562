// MFMA GEMM / implicit-GEMM 3x3 convolution for gfx950 (MI355X).
//
//   C[M][N] = epilogue( A[M][K] · W[N][K]^T ),  16-bit inputs (fp16 | bf16), fp32 accumulate.
//
// Replaces every nn.Linear / nn.Conv2d(3x3) on the SEED-X hot path (see include/seedx_hip.h for the
// reference call sites). Design:
//   * block = 4 or 8 waves (8 = 2 per SIMD) laid out WM x WN over a BM x BN x 64 tile; each wave owns (BM/WM)x(BN/WN) as
//     16x16x32 MFMA fragments. Tile menu: 128x128 (2x2), 128x80 (4x1), 64x128, 64x64 with 4 waves; 256x256 (2x4), 256x320
//     (2x4), 256x160 (4x2) with 8 waves. N of the SDXL UNet is k*320, so the x320 / x160 / x80 tiles split it without a
//     ragged last tile and land on whole rounds of 256 tiles; pick_tile() takes the argmin of a fitted cost model
//   * A and W tiles are DMA'd HBM→LDS with `buffer_load_dwordx4 … lds` (no VGPR round trip); rows
//     beyond M / N and zero-padding taps of the convolution use the buffer descriptor's range check
//     (offset >= num_records returns 0), so there is no edge code in the main loop
//   * LDS rows are 128 B (64 k-elements); the 16-B chunk index is XOR-swizzled with (row & 7). The DMA
//     destination is lane-linear, so the swizzle is applied to the per-lane SOURCE address and again on
//     the ds_read_b128 side (same involution) → conflict-free fragment reads
//   * NSTAGE-deep LDS ring (2 or 3), ONE raw s_barrier per k-tile; DMA of later tiles stays in flight
//     across the barrier and is retired with a COUNTED `s_waitcnt vmcnt(N)` (never a drain in steady state)
//   * operands are swapped in the MFMA (D = Wfrag · Afrag^T) so a lane ends up with 4 CONSECUTIVE output
//     columns of one row → vector bias/residual loads and 8/16-byte stores in the fused epilogue
//   * 1-D grid; block b runs on XCD b % 8, each XCD owns a compact rectangle of the tile grid and walks it in groups of
//     8 tile-rows, so its private 4-MB L2 serves most operand re-reads (small grids: bijective XCD remap)
#include "gemm_common.h"

namespace sxk_gemm {

template <typename TT, int BM, int BN, int WM, int WN, int NSTAGE, int AMODE>
__global__ __launch_bounds__(WM * WN * 64) void gemm_kernel(const GemmP p) {
#if defined(__HIP_DEVICE_COMPILE__)  // body uses gfx950-only builtins (LDS-DMA, MFMA); the host pass only needs the stub
  typedef typename TT::vec8 vec8;
  constexpr int NW = WM * WN;                     // waves per block: 4 (256 threads) or 8 (512 threads, 2 per SIMD)
  static_assert(NW == 4 || NW == 8, "4 or 8 waves per block");
  constexpr int TM = BM / WM, TN = BN / WN;       // wave tile
  constexpr int FM = TM / 16, FN = TN / 16;       // 16x16 fragments per wave along m / n
  constexpr bool kGlu = (FN % 2 == 0);            // GLU pairs fragments (i, i+1): tiles with an odd FN (256x320, 128x80) never run it,
                                                  // so their epilogue carries none of its code or registers
  constexpr int BNP = (BN + 8 * NW - 1) / (8 * NW) * (8 * NW);  // W rows staged per tile (every wave issues the same count)
  constexpr int A_PER_WAVE = BM / (8 * NW);       // 1-KiB DMA slots (8 rows x 128 B) per wave per k-tile
  constexpr int B_PER_WAVE = BNP / (8 * NW);
  constexpr int LOADS = A_PER_WAVE + B_PER_WAVE;  // DMA instructions per wave per k-tile (vmcnt unit)
  constexpr int A_BYTES = BM * 128, B_BYTES = BNP * 128, STAGE = A_BYTES + B_BYTES;
  static_assert(LOADS * (NSTAGE - 1) <= 63, "vmcnt range");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  unsigned long long t_start = 0, t_first = 0, t_main = 0;
  if (p.dbg) t_start = __builtin_amdgcn_s_memtime();

  int tile_m, tile_n;
  if (!tile_coords(p, blockIdx.x, gridDim.x, tile_m, tile_n)) return;  // padding block of an uneven split (exits before any barrier)
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)p.a_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (int)p.w_bytes, 0x00020000);

  // ---- per-lane DMA source descriptors (fixed over the k loop) ---------------------------------
  const int rl = lane >> 3;                         // row within the 8-row slot
  const unsigned gchunk = ((lane & 7) ^ rl) << 4;   // swizzled 16-B chunk this lane fetches
  unsigned a_off[A_PER_WAVE];                       // LINEAR: byte offset of the row; CONV: unused
  int a_b[A_PER_WAVE], a_y[A_PER_WAVE], a_x[A_PER_WAVE];
  unsigned w_off[B_PER_WAVE];
#pragma unroll
  for (int i = 0; i < A_PER_WAVE; ++i) {
    const int row = m0 + (wave * A_PER_WAVE + i) * 8 + rl;
    if (AMODE == SX_A_LINEAR) {
      a_off[i] = (row < p.M) ? (unsigned)row * (unsigned)p.K * 2u + gchunk : 0x80000000u;
      a_b[i] = a_y[i] = a_x[i] = 0;
    } else {
      const int hw = p.Hout * p.Wout;
      const int b = row / hw, rem = row - b * hw;
      const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
      a_b[i] = (row < p.M) ? b * p.Hin : -(1 << 28);  // invalid rows fail the range test below
      a_y[i] = oy * p.stride - p.pad;
      a_x[i] = ox * p.stride - p.pad;
      a_off[i] = 0;
    }
  }
#pragma unroll
  for (int i = 0; i < B_PER_WAVE; ++i) {
    const int lrow = (wave * B_PER_WAVE + i) * 8 + rl;   // rows >= BN are padding: fetched as zeros, never read
    const int row = n0 + lrow;
    w_off[i] = (row < p.N && lrow < BN) ? (unsigned)row * (unsigned)p.Kw * 2u + gchunk : 0x80000000u;
  }
  const int Hv = p.upsample ? 2 * p.Hin : p.Hin, Wv = p.upsample ? 2 * p.Win : p.Win;
  const int cpt = (AMODE == SX_A_CONV3X3) ? p.Cin / 64 : 1;  // k-tiles per filter tap
  const int nkw = p.Kw / 64;                                 // k-tiles of W (a_planes = 2: half of the loop's, walked twice)

  auto stage = [&](int buf, int kt) {
    unsigned char* sA = smem + buf * STAGE;
    unsigned char* sB = sA + A_BYTES;
    int dy = 0, dx = 0, cc = 0;
    if (AMODE == SX_A_CONV3X3) {
      const int tap = kt / cpt;
      cc = kt - tap * cpt;
      dy = tap / 3;
      dx = tap - dy * 3;
    }
#pragma unroll
    for (int i = 0; i < A_PER_WAVE; ++i) {
      unsigned voff;
      if (AMODE == SX_A_LINEAR) {
        voff = a_off[i] + (unsigned)kt * 128u;
      } else {
        int iy = a_y[i] + dy, ix = a_x[i] + dx;
        const bool ok = (a_b[i] >= 0) && iy >= 0 && iy < Hv && ix >= 0 && ix < Wv;
        if (p.upsample) { iy >>= 1; ix >>= 1; }
        voff = ok ? (unsigned)(((a_b[i] + iy) * p.Win + ix) * p.Cin) * 2u + (unsigned)cc * 128u + gchunk
                  : 0x80000000u;
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, SX_LDS_PTR(sA + (wave * A_PER_WAVE + i) * 1024), 16, voff, 0, 0,
                                               0);
    }
#pragma unroll
    for (int i = 0; i < B_PER_WAVE; ++i) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, SX_LDS_PTR(sB + (wave * B_PER_WAVE + i) * 1024), 16,
                                               w_off[i] + (unsigned)(kt >= nkw ? kt - nkw : kt) * 128u, 0, 0, 0);
    }
  };

  f32x4_t acc[FN][FM];
  if (p.res_init) {
    // fp32 residual (no activation, no GLU) is the accumulators' initial value — same rounding order in every tile
    // config (gemm_pp.hip does the same), and the epilogue keeps no loads behind its stores
#pragma unroll
    for (int j = 0; j < FM; ++j) {
      const int m = m0 + wm * TM + j * 16 + (lane & 15);
      const int mc = m < p.M ? m : p.M - 1;
      const float* rr = p.residual + (size_t)(p.res_mod ? (mc % p.res_mod) : mc) * p.ldr;
#pragma unroll
      for (int i = 0; i < FN; ++i) {
        int c = n0 + wn * TN + i * 16 + (lane >> 4) * 4;
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

  const int nkt = p.K / 64;
  constexpr int D = NSTAGE - 1;  // prefetch distance in k-tiles
#pragma unroll
  for (int s = 0; s < D; ++s)
    if (s < nkt) stage(s, s);
  int cur = 0, nxt = D % NSTAGE;  // ring slots of tile kt and tile kt + D
  for (int kt = 0; kt < nkt; ++kt) {
    // tile kt must have landed (this wave's share): allow the younger (D-1) tiles to stay in flight
    if (D >= 2 && kt + 1 < nkt) wait_vmcnt<LOADS*(D - 1)>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();  // everyone's share landed AND everyone finished reading ring slot `nxt`
    if (p.dbg && kt == 0) t_first = __builtin_amdgcn_s_memtime();
    if (kt + D < nkt) stage(nxt, kt + D);
    const unsigned char* sA = smem + cur * STAGE + (wm * TM) * 128;
    const unsigned char* sB = smem + cur * STAGE + A_BYTES + (wn * TN) * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      vec8 af[FM], wf[FN];
#pragma unroll
      for (int j = 0; j < FM; ++j) af[j] = *(const vec8*)(sA + j * 2048 + frag_row + frag_sw[ks]);
#pragma unroll
      for (int i = 0; i < FN; ++i) wf[i] = *(const vec8*)(sB + i * 2048 + frag_row + frag_sw[ks]);
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = TT::mfma16(wf[i], af[j], acc[i][j]);
    }
    cur = (cur + 1 == NSTAGE) ? 0 : cur + 1;
    nxt = (nxt + 1 == NSTAGE) ? 0 : nxt + 1;
  }

  if (p.dbg) t_main = __builtin_amdgcn_s_memtime();
  // ---- fused epilogue: lane holds C[m][n .. n+3], m = ..+(lane&15), n = ..+(lane>>4)*4 ----------
  // Loads are issued unconditionally on clamped addresses and batched per phase (bias once, then per
  // m-fragment: bias2d + residual for all n-fragments) so they overlap instead of serialising on vmcnt(0).
  const int lq = (lane >> 4) * 4;
  int ncol[FN], nout[FN];
  bool nok[FN];
  f32x4_t bv[FN];
#pragma unroll
  for (int i = 0; i < FN; ++i) {
    const int nb = n0 + wn * TN + i * 16;
    ncol[i] = nb + lq;
    nout[i] = (kGlu && p.glu) ? (nb >> 1) + lq : ncol[i];
    nok[i] = ncol[i] < p.N && nout[i] < p.n_valid && !((kGlu && p.glu) && (i & 1));
    if (ncol[i] >= p.N) ncol[i] = p.N - 4;
    if (nout[i] + 4 > p.ldc) nout[i] = 0;  // clamped lanes never store
    bv[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }
  if (p.bias) {
#pragma unroll
    for (int i = 0; i < FN; ++i) bv[i] = *(const f32x4_t*)(p.bias + ncol[i]);
  }
  const bool wide = p.out_dtype != SX_F32 && !(kGlu && p.glu) && (p.ldc & 7) == 0 && (((size_t)p.C) & 15) == 0;
  bool pair_ok[(FN + 1) / 2];
#pragma unroll
  for (int i = 0; i < (FN + 1) / 2; ++i) {
    const int lim = p.n_valid < p.N ? p.n_valid : p.N;
    pair_ok[i] = (2 * i + 1 < FN) && (n0 + wn * TN + i * 32 + 32 <= lim);  // wave-uniform: the whole 32-col pair is stored
  }
#pragma unroll
  for (int j = 0; j < FM; ++j) {
    const int m = m0 + wm * TM + j * 16 + (lane & 15);
    const bool mok = m < p.M;
    const int mc = mok ? m : p.M - 1;
    f32x4_t v[FN], rv[FN];
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      v[i] = acc[i][j] + bv[i];
      rv[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    if (p.bias2d) {
      const float* b2 = p.bias2d + (size_t)(mc / p.bias2d_rows) * p.ldb2;
#pragma unroll
      for (int i = 0; i < FN; ++i) v[i] += *(const f32x4_t*)(b2 + ncol[i]);
    }
    if (p.residual && !p.res_init) {
      const float* rr = p.residual + (size_t)(p.res_mod ? (mc % p.res_mod) : mc) * p.ldr;
#pragma unroll
      for (int i = 0; i < FN; ++i) {
        const int c = (nout[i] + 4 <= p.ldr) ? nout[i] : 0;
        rv[i] = *(const f32x4_t*)(rr + c);
      }
    }
    if (wide) {
      // 16-bit output, no GLU: two neighbouring n-fragments hold cols [nb, nb+16) and [nb+16, nb+32) as 4 per lane.
      // v_permlane16_swap trades the odd 16-lane rows of fragment i with the even rows of fragment i+1, after which a
      // lane owns 8 CONSECUTIVE columns → one 16-B store instead of two 8-B stores (the epilogue is store-issue bound)
#pragma unroll
      for (int i = 0; i + 1 < FN; i += 2) {
        if (!pair_ok[i >> 1]) continue;
        u32x2_t o[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          f32x4_t x = v[i + h];
          if (p.act == SX_ACT_GELU) {
            const f32x2_t g0 = gelu_erf2((f32x2_t){x[0], x[1]}), g1 = gelu_erf2((f32x2_t){x[2], x[3]});
            x = (f32x4_t){g0[0], g0[1], g1[0], g1[1]};
          } else if (p.act != SX_ACT_NONE) {
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = apply_act(x[r], p.act);
          }
          x += rv[i + h];
          if (p.out_dtype == SX_BF16) {
            o[h][0] = pack2<BF16>(x[0], x[1]);
            o[h][1] = pack2<BF16>(x[2], x[3]);
          } else {
            o[h][0] = pack2<F16>(x[0], x[1]);
            o[h][1] = pack2<F16>(x[2], x[3]);
          }
        }
        const auto s0 = __builtin_amdgcn_permlane16_swap(o[0][0], o[1][0], false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(o[0][1], o[1][1], false, false);
        if (!mok) continue;
        const u32x4_t w4 = {s0[0], s1[0], s0[1], s1[1]};
        const int q = lane >> 4;
        const int col = n0 + wn * TN + i * 16 + (q & 1) * 16 + (q >> 1) * 8;
        *(u32x4_t*)((unsigned short*)p.C + (size_t)m * p.ldc + col) = w4;
      }
    }
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      if (wide && pair_ok[i >> 1]) continue;  // stored by the 16-B path above
      if ((kGlu && p.glu)) {
        if (i & 1) continue;
        const f32x4_t g = v[(i + 1) < FN ? (i + 1) : i];
        if (p.act == SX_ACT_GELU) {   // GEGLU (SDXL feed-forward): packed-fp32 GELU, two gates per VALU issue
          const f32x2_t g0 = gelu_erf2((f32x2_t){g[0], g[1]}), g1 = gelu_erf2((f32x2_t){g[2], g[3]});
          v[i][0] *= g0[0]; v[i][1] *= g0[1]; v[i][2] *= g1[0]; v[i][3] *= g1[1];
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[i][r] = v[i][r] * apply_act(g[r], p.act);
        }
      } else if (p.act == SX_ACT_GELU) {
        const f32x2_t g0 = gelu_erf2((f32x2_t){v[i][0], v[i][1]}), g1 = gelu_erf2((f32x2_t){v[i][2], v[i][3]});
        v[i] = (f32x4_t){g0[0], g0[1], g1[0], g1[1]};
      } else if (p.act != SX_ACT_NONE) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[i][r] = apply_act(v[i][r], p.act);
      }
      const f32x4_t o4 = v[i] + rv[i];
      if (!(mok && nok[i])) continue;
      if (p.out_dtype == SX_F32) {
        *(f32x4_t*)((float*)p.C + (size_t)m * p.ldc + nout[i]) = o4;
      } else {
        u32x2_t o;
        if (p.out_dtype == SX_BF16) {
          o[0] = pack2<BF16>(o4[0], o4[1]);
          o[1] = pack2<BF16>(o4[2], o4[3]);
        } else {
          o[0] = pack2<F16>(o4[0], o4[1]);
          o[1] = pack2<F16>(o4[2], o4[3]);
        }
        *(u32x2_t*)((unsigned short*)p.C + (size_t)m * p.ldc + nout[i]) = o;
      }
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

unsigned long long* g_dbg = nullptr;   // tuning hook: timestamp buffer for the next launches (sx_gemm_debug_stamps)
int g_gm = 0;      // tuning hook: tile-rows per traversal group (0 = default)
int g_xcd_2d = 1;  // 2-D XCD tile partition on/off (tuning hook)

template <typename TT, int BM, int BN, int WM, int WN, int NSTAGE>
int launch_cfg(const GemmP& p0, int a_mode, hipStream_t st) {
  GemmP p = p0;
  p.dbg = g_dbg;
  const int grid = plan_grid(p, BM, BN, g_xcd_2d, g_gm);
  constexpr int NW = WM * WN;
  const size_t lds = (size_t)NSTAGE * (BM + (BN + 8 * NW - 1) / (8 * NW) * (8 * NW)) * 128;
  // the LDS opt-in is a per-device function attribute: cache it per device (a VAE on a second GPU needs its own call)
  auto reserve = [&](const void* k, hipError_t* attr, bool* done) -> hipError_t {
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev &= 15;
    if (!done[dev]) {
      attr[dev] = lds > 65536 ? hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) : hipSuccess;
      done[dev] = true;
    }
    return attr[dev];
  };
  if (a_mode == SX_A_LINEAR) {
    auto k = gemm_kernel<TT, BM, BN, WM, WN, NSTAGE, SX_A_LINEAR>;
    static hipError_t attr[16];
    static bool done[16];
    const hipError_t e = reserve((const void*)k, attr, done);
    SX_CHECK(e == hipSuccess, "sx_gemm: cannot reserve %zu B of LDS for the %dx%d tile: %s", lds, BM, BN, hipGetErrorString(e));
    hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), lds, st, p);
  } else {
    auto k = gemm_kernel<TT, BM, BN, WM, WN, NSTAGE, SX_A_CONV3X3>;
    static hipError_t attr[16];
    static bool done[16];
    const hipError_t e = reserve((const void*)k, attr, done);
    SX_CHECK(e == hipSuccess, "sx_gemm: cannot reserve %zu B of LDS for the %dx%d tile: %s", lds, BM, BN, hipGetErrorString(e));
    hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), lds, st, p);
  }
  SX_HIP_LAUNCH_CHECK();
  return SX_OK;
}

// tile menu + cost model. One launch runs ceil(tiles / slots) rounds over the chip (slots = 256 CUs x co-resident blocks) and
// a round costs a + b*K microseconds (prologue/epilogue + per-k-tile time). The constants are a least-squares fit of the
// MI355X sweep tools/bench_tile_model.py (profiles/r1_tile_model.jsonl): every tile config forced over the UNet linear and
// conv shapes at CFG batch 2..32, the ViT-G and the Llama shapes. Picking argmin of the model is within 0.2 % (linear) /
// 0.1 % (conv) of the per-shape best config on the re-run of that sweep with the final kernels (worst shape 9 %); the N = k*320 channel counts of the SDXL UNet are why the
// 256x320 / 256x160 tiles exist (they split N without a ragged last tile and land on whole rounds of 256 tiles).
struct TileCfg { int bm, bn; bool glu_ok; int slots; float a_lin, b_lin, a_conv, b_conv; };
static const int kNumTiles = 9;   // 0..6 lock-step kernels (this file), 7 / 8 ping-pong 256x256 / 256x320 (gemm_pp.hip)
static const TileCfg kTiles[kNumTiles] = {
    // round 3 fit (tools/lab/gemm_lab model → tools/fit_tile_model.py, profiles/r3_tile_model.jsonl): argmin of the model
    // is within 0.2 % of the per-shape best config summed over the sweep (worst shape 8 %)
    {128, 128, true, 512, 7.85f, 0.01202f, 15.26f, 0.01513f},  {128, 80, false, 256, 3.45f, 0.00766f, 3.59f, 0.01135f},
    {64, 128, true, 256, 1.37f, 0.00479f, 1.63f, 0.00546f},    {64, 64, true, 256, 0.33f, 0.00301f, 0.82f, 0.00313f},
    {256, 256, true, 256, 11.34f, 0.02161f, 13.19f, 0.02365f}, {256, 320, false, 256, 17.95f, 0.02566f, 18.01f, 0.02985f},
    {256, 160, false, 256, 8.13f, 0.01579f, 9.06f, 0.01897f},  {256, 256, true, 256, 10.56f, 0.01592f, 12.81f, 0.01840f},
    {256, 320, false, 256, 13.32f, 0.01928f, 15.12f, 0.02173f}};

inline int pick_tile(int M, int N, int K, bool glu, bool conv, int force, unsigned allow = 0x7f) {
  if (force >= 0 && force < kNumTiles && (!glu || kTiles[force].glu_ok)) return force;   // (force 9 = strip kernel: decided by the caller)
  int best = 2;
  float best_t = 1e30f;
  for (int c = 0; c < kNumTiles; ++c) {
    const TileCfg& t = kTiles[c];
    if (!((allow >> c) & 1u)) continue;
    if (glu && !t.glu_ok) continue;
    const long tiles = (long)((M + t.bm - 1) / t.bm) * ((N + t.bn - 1) / t.bn);
    const long rounds = (tiles + t.slots - 1) / t.slots;
    const float us = (float)rounds * (conv ? t.a_conv + t.b_conv * (float)K : t.a_lin + t.b_lin * (float)K);
    if (us < best_t) { best_t = us; best = c; }
  }
  return best;
}

}  // namespace sxk_gemm
using namespace sxk_gemm;

static int g_force_tile = -1;
// The strip kernel is NOT part of the automatic choice: it is correct (bit-identical C / x16, reproducible row sums) and slower than the
// 256x320 ping-pong producer on every production shape (profiles/r6_ab_experiments.md §1: its 128x256 sub-tiles need twice the operand
// bytes per flop and the CU's LDS-DMA path bounds it). sx_gemm_force_tile(9) / (501) select it for the lab and the tests.
static int g_use_strip = 0;
namespace sxk_gemm { int g_use_pp = 1; }  // 0 = lock-step kernels only (A/B hook, sx_gemm_force_tile(200))
extern "C" int sx_gemm_pick_tile(int M, int N, int K, int glu, int conv) {  // host-only: which tile config sx_gemm would use
  return sxk_gemm::pick_tile(M, N, K, glu != 0, conv != 0, -1, 0x1ff);
}

extern "C" int sx_gemm_debug_stamps(void* buf) {   // tuning hook: device buffer of 4 x uint64 per block, or NULL to switch off
  sxk_gemm::g_dbg = (unsigned long long*)buf;
  return SX_OK;
}

extern "C" int sx_gemm_force_tile(int cfg) {  // tuning / test hook: -1 = automatic; 100/101 = 2-D XCD partition off/on
  if (cfg == 100 || cfg == 101) { sxk_gemm::g_xcd_2d = cfg - 100; return SX_OK; }
  if (cfg >= 300 && cfg <= 364) { sxk_gemm::g_gm = cfg - 300; return SX_OK; }
  if (cfg >= 400 && cfg <= 409) { sxk_gemm::g_pp_variant = cfg - 400; return SX_OK; }
  if (cfg == 200 || cfg == 201) { sxk_gemm::g_use_pp = cfg - 200; return SX_OK; }
  if (cfg == 500 || cfg == 501) { g_use_strip = cfg - 500; return SX_OK; }
  if (cfg >= 600 && cfg <= 663) { sxk_gemm::g_tune = cfg - 600; return SX_OK; }
  g_force_tile = cfg;
  return SX_OK;
}

static int gemm_impl(const sx_gemm_args* a, double* gn_stats, int gn_groups, int gn_rows, int* gn_fused, const sx_gemm_ln_args* ln,
                     void* stream);

extern "C" int sx_gemm(const sx_gemm_args* a, void* stream) { return gemm_impl(a, nullptr, 0, 0, nullptr, nullptr, stream); }

// sx_gemm with a LayerNorm folded into the two GEMMs around it (include/seedx_hip.h sx_gemm_ln_args). Ping-pong tiles only: the call
// fails when the cost model would run this shape on a lock-step tile — ask sx_gemm_pick_tile first and keep the separate
// sx_layernorm launch for those shapes.
extern "C" int sx_gemm_ln(const sx_gemm_args* a, const sx_gemm_ln_args* ln, void* stream) {
  SX_CHECK(ln && (ln->row_stats_in || ln->row_stats_out), "sx_gemm_ln: neither a producer nor a consumer role");
  return gemm_impl(a, nullptr, 0, 0, nullptr, ln, stream);
}

// sx_gemm + GroupNorm statistics of the output it stores (the statistics pass of the NEXT sx_groupnorm, fused into this launch's
// epilogue): stats[row / rows_per_sample][column / (N / groups)][2] += (sum, sum of squares), fp64, by atomics — `stats` must be
// zero (or hold the partial sums of other launches) beforehand. Only the ping-pong tiles carry the fused epilogue: *fused (host)
// reports whether THIS launch accumulated; if 0 the GEMM ran unchanged and the caller runs sx_groupnorm's own statistics pass.
extern "C" int sx_gemm_gn(const sx_gemm_args* a, double* stats, int groups, int rows_per_sample, int* fused, void* stream) {
  SX_CHECK(stats && fused && groups > 0 && rows_per_sample > 0, "sx_gemm_gn: bad statistics arguments");
  return gemm_impl(a, stats, groups, rows_per_sample, fused, nullptr, stream);
}

static int gemm_impl(const sx_gemm_args* a, double* gn_stats, int gn_groups, int gn_rows, int* gn_fused, const sx_gemm_ln_args* ln,
                     void* stream) {
  SX_CHECK(a && a->A && a->W && a->C, "sx_gemm: null pointer");
  if (gn_fused) *gn_fused = 0;
  SX_CHECK(a->dtype == SX_F16 || a->dtype == SX_BF16, "sx_gemm: dtype must be f16/bf16");
  SX_CHECK(a->out_dtype >= SX_F16 && a->out_dtype <= SX_F32, "sx_gemm: bad out_dtype");
  SX_CHECK(a->M > 0 && a->N > 0 && a->K > 0, "sx_gemm: empty problem M=%d N=%d K=%d", a->M, a->N, a->K);
  SX_CHECK(a->K % 64 == 0, "sx_gemm: K=%d must be a multiple of 64 (pad the operand)", a->K);
  SX_CHECK(a->N % 16 == 0, "sx_gemm: N=%d must be a multiple of 16 (pad the weight)", a->N);
  SX_CHECK(!a->glu || a->N % 32 == 0, "sx_gemm: glu needs N %% 32 == 0");
  const int n_out = a->glu ? a->N / 2 : a->N;
  SX_CHECK(a->n_valid >= 0 && a->n_valid % 4 == 0 && a->n_valid <= n_out, "sx_gemm: n_valid=%d", a->n_valid);
  SX_CHECK(a->ldc >= (a->n_valid > 0 ? a->n_valid : n_out) && a->ldc % 4 == 0, "sx_gemm: ldc=%d invalid for n_out=%d",
           a->ldc, n_out);
  SX_CHECK(!a->residual || (a->ldr >= (a->n_valid > 0 ? a->n_valid : n_out) && a->ldr % 4 == 0), "sx_gemm: ldr=%d invalid", a->ldr);
  SX_CHECK(!a->bias2d || a->bias2d_rows > 0, "sx_gemm: bias2d_rows must be > 0");
  GemmP p;
  memset(&p, 0, sizeof(p));
  p.A = a->A; p.W = a->W; p.C = a->C; p.bias = a->bias; p.bias2d = a->bias2d; p.residual = a->residual;
  // a_planes = 2: A carries the hi and lo planes of an fp32-grade activation side by side; the kernels walk 2K with W wrapped at K
  SX_CHECK(a->a_planes >= 0 && a->a_planes <= 2, "sx_gemm: a_planes=%d", a->a_planes);
  const int planes = a->a_planes == 2 ? 2 : 1;
  SX_CHECK(planes == 1 || (a->a_mode == SX_A_LINEAR && !ln), "sx_gemm: a_planes = 2 is for plain linear GEMMs");
  const int Kk = a->K * planes;
  p.M = a->M; p.N = a->N; p.K = Kk; p.Kw = a->K; p.ldc = a->ldc; p.ldr = a->ldr; p.res_mod = a->res_mod;
  p.n_valid = a->n_valid > 0 ? a->n_valid : n_out;
  p.bias2d_rows = a->bias2d_rows; p.out_dtype = a->out_dtype; p.act = a->act; p.glu = a->glu;
  p.ldb2 = a->ld_bias2d > 0 ? a->ld_bias2d : a->N;
  p.pad = 1;
  uint64_t a_bytes;
  if (a->a_mode == SX_A_CONV3X3) {
    SX_CHECK(a->Cin % 64 == 0 && a->K == 9 * a->Cin, "sx_gemm conv: Cin=%d K=%d", a->Cin, a->K);
    SX_CHECK(a->stride == 1 || a->stride == 2, "sx_gemm conv: stride");
    SX_CHECK(a->M == a->B * a->Hout * a->Wout, "sx_gemm conv: M != B*Hout*Wout");
    const int hv = a->upsample ? 2 * a->Hin : a->Hin, wv = a->upsample ? 2 * a->Win : a->Win;
    SX_CHECK(a->pad_mode == 0 || a->pad_mode == 1, "sx_gemm conv: pad_mode");
    const int padsum = a->pad_mode ? 1 : 2;
    SX_CHECK(a->Hout == (hv + padsum - 3) / a->stride + 1 && a->Wout == (wv + padsum - 3) / a->stride + 1,
             "sx_gemm conv: output geometry mismatch");
    p.pad = a->pad_mode ? 0 : 1;
    p.Hin = a->Hin; p.Win = a->Win; p.Cin = a->Cin; p.Hout = a->Hout; p.Wout = a->Wout;
    p.stride = a->stride; p.upsample = a->upsample;
    a_bytes = (uint64_t)a->B * a->Hin * a->Win * a->Cin * 2;
  } else {
    SX_CHECK(a->a_mode == SX_A_LINEAR, "sx_gemm: bad a_mode");
    a_bytes = (uint64_t)a->M * Kk * 2;
  }
  const uint64_t w_bytes = (uint64_t)a->N * a->K * 2;
  SX_CHECK(a_bytes < 0x7fffffffull && w_bytes < 0x7fffffffull, "sx_gemm: operand exceeds 2 GiB descriptor range");
  p.a_bytes = (unsigned)a_bytes;
  p.w_bytes = (unsigned)w_bytes;
  hipStream_t st = (hipStream_t)stream;
  p.res_init = (p.residual && p.act == SX_ACT_NONE && !p.glu) ? 1 : 0;
  // tile configs 0..6: lock-step kernels (this file); 7 / 8: ping-pong 256x256 / 256x320 (gemm_pp.hip). The cost model
  // ranks the lock-step menu; where it picks a 256-row 8-wave tile, the ping-pong kernel of the same shape runs instead
  // when its epilogue combination exists (forced 4 / 5 keep the lock-step kernels for A/B runs).
  // tile configs 0..6: lock-step kernels (this file); 7 / 8: ping-pong 256x256 / 256x320 (gemm_pp.hip), offered to the cost
  // model when their epilogue combination is instantiated
  if (ln) {
    SX_CHECK(a->a_mode == SX_A_LINEAR && !gn_stats, "sx_gemm_ln: linear GEMMs only");
    SX_CHECK(!(ln->row_stats_in && ln->row_stats_out), "sx_gemm_ln: a launch is a producer or a consumer, not both");
    if (ln->row_stats_out) {
      SX_CHECK(ln->x16_out && a->out_dtype == SX_F32 && !a->glu && a->act == SX_ACT_NONE && p.n_valid == a->N,
               "sx_gemm_ln producer: needs x16_out and a plain fp32 output of all N columns");
      SX_CHECK(ln->ld_x16 >= a->N && ln->ld_x16 % 4 == 0 && (((size_t)ln->x16_out) & 7) == 0, "sx_gemm_ln producer: ld_x16=%d", ln->ld_x16);
      p.ln_x16 = ln->x16_out; p.ln_ldx = ln->ld_x16; p.ln_out = ln->row_stats_out;
    } else {
      SX_CHECK(ln->colsum && ln->dim == a->K && ln->eps > 0.f, "sx_gemm_ln consumer: colsum / dim (= K) / eps");
      SX_CHECK(a->out_dtype == a->dtype && !a->residual && !a->bias2d, "sx_gemm_ln consumer: 16-bit output, no residual / bias2d");
      p.ln_in = ln->row_stats_in; p.ln_cs = ln->colsum; p.ln_eps = ln->eps; p.ln_inv_dim = 1.0f / (float)ln->dim;
    }
  }
  unsigned allow = 0x7f;
  if (ln) allow = 0;            // ping-pong tiles or nothing
  if (g_use_pp || g_force_tile == 7) allow |= pp_supported(p, a->dtype, 256, a->a_mode) ? 0x80u : 0u;
  if (g_use_pp || g_force_tile == 8) allow |= pp_supported(p, a->dtype, 320, a->a_mode) ? 0x100u : 0u;
  SX_CHECK(!(g_force_tile == 7 || g_force_tile == 8) || ((allow >> g_force_tile) & 1u),
           "sx_gemm: forced ping-pong tile has no kernel for this epilogue");
  // LayerNorm producers (fp32 residual in, fp32 + 16-bit out, row sums): the persistent strip kernel when the launch has at least
  // 3/4 of a strip per CU (fewer strips leave CUs idle that the one-tile-per-workgroup kernels would use) — forced tile 9 = always
  if (p.ln_out && (g_force_tile == 9 || (g_use_strip && g_force_tile < 0 && a->M / 128 >= 192)) && strip_supported(p, a->a_mode))
    return launch_strip(p, a->dtype, st);
  SX_CHECK(g_force_tile != 9, "sx_gemm: forced strip kernel does not support this launch");
  if (ln) {
    // the fold exists on the tiles the cost model gives this shape without it, or not at all (no silent change of tile)
    const int plain = pick_tile(a->M, a->N, Kk, a->glu != 0, false, g_force_tile, 0x1ff);
    SX_CHECK((plain == 7 || plain == 8) && ((allow >> plain) & 1u),
             "sx_gemm_ln: M=%d N=%d K=%d does not run on a ping-pong tile with this epilogue (tile %d): keep sx_layernorm for it", a->M,
             a->N, a->K, plain);
    allow = 1u << plain;
  }
  const int cfg = pick_tile(a->M, a->N, Kk, a->glu != 0, a->a_mode == SX_A_CONV3X3, g_force_tile, allow);
  if (cfg == 7 || cfg == 8) {
    // fused GroupNorm statistics: fp32 output of all N columns, whole 256-row tiles inside one sample, even channels per group
    if (gn_stats && a->out_dtype == SX_F32 && !a->glu && a->act == SX_ACT_NONE && p.n_valid == a->N && a->N % gn_groups == 0 &&
        (a->N / gn_groups) % 2 == 0 && gn_rows % 256 == 0 && a->M % gn_rows == 0) {
      p.gn_stats = gn_stats; p.gn_groups = gn_groups; p.gn_cpg = a->N / gn_groups; p.gn_rows = gn_rows;
      *gn_fused = 1;
    }
    return launch_pp(p, a->dtype, cfg == 7 ? 256 : 320, a->a_mode, st);
  }
#define SX_GEMM_DISPATCH(TT)                                                  \
  switch (cfg) {                                                              \
    case 0: return launch_cfg<TT, 128, 128, 2, 2, 2>(p, a->a_mode, st);       \
    case 1: return launch_cfg<TT, 128, 80, 4, 1, 3>(p, a->a_mode, st);        \
    case 2: return launch_cfg<TT, 64, 128, 2, 2, 3>(p, a->a_mode, st);        \
    case 3: return launch_cfg<TT, 64, 64, 2, 2, 3>(p, a->a_mode, st);         \
    case 4: return launch_cfg<TT, 256, 256, 2, 4, 2>(p, a->a_mode, st);       \
    case 5: return launch_cfg<TT, 256, 320, 2, 4, 2>(p, a->a_mode, st);       \
    default: return launch_cfg<TT, 256, 160, 4, 2, 2>(p, a->a_mode, st);      \
  }
  if (a->dtype == SX_BF16) { SX_GEMM_DISPATCH(BF16) } else { SX_GEMM_DISPATCH(F16) }
#undef SX_GEMM_DISPATCH
}
