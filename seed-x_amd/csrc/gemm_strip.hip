// Persistent "strip" GEMM for the fp32-residual / LayerNorm-producer launches of the UNet's transformer blocks (gfx950, MI355X):
//
//     C[M][N] (fp32) = residual[M][N] (fp32) + A[M][K] · W[N][K]^T + bias[N],     x16[M][N] = round16(C),   rows[M] = (Σ C, Σ C²)
//
// (attention out-projections and ff.net.2 of diffusers' BasicTransformerBlock [ext] as the t2i loop calls them,
// pipeline_stable_diffusion_xl_t2i_edit.py:915-922; same contract as gemm_pp.hip's LN = 2 producer epilogue, include/seedx_hip.h
// sx_gemm_ln). These launches move 12 B per output element for 2·K flops: at K = 1280 their roofline is HBM, and the one-tile-per-
// workgroup kernels run three serial phases per tile on a CU that holds ONE workgroup — residual pre-load (HBM), main loop (MFMA,
// HBM idle), stores (HBM). Here the three overlap:
//
//   * one workgroup per CU walks whole 128-row STRIPS of the output (grid = min(strips, CUs)), 256 columns (a "sub-tile") at a time;
//   * TWO accumulator sets of 64 registers: while sub-tile t accumulates into set t & 1, the other set is — fragment by fragment, one
//     16x16 fragment per barrier interval of the first 8 k-tiles — turned into sub-tile t-1's output (bias add, fp32 store, 16-bit
//     copy, row sums) and immediately re-loaded with sub-tile t+1's residual. Loads, stores and the operand LDS-DMA share vmcnt in
//     issue order, every access is a buffer instruction whose descriptor is the NULL descriptor when the neighbour sub-tile does not
//     exist, so each barrier interval issues a compile-time number of VM operations and the counted waits stay exact;
//   * the operand ring (3 stages x (128 A rows + 256 W rows) x 128 B) never drains: the DMA stream of global k-tile q + 2 is issued
//     during k-tile q across sub-tile and strip boundaries;
//   * a strip covers all N columns, so the rows' (Σ, Σ²) are complete inside one workgroup: lane partials → two xor shuffles → the
//     four column waves meet in LDS in a fixed order → ONE plain 16-byte store per row. No atomics: the statistics (and everything
//     the LayerNorm-fold consumers compute from them) are bit-reproducible from run to run.
//
// Schedule inside a k-tile: the 8-wave ping-pong of gemm_pp.hip (two groups of four waves one barrier interval apart; a group's
// load segment {fragment ds_reads, 3 DMA issues, its epilogue piece, counted vmcnt, lgkmcnt(0)} runs under the other group's 16
// MFMAs), two 32-deep phases per 64-deep k-tile; wave (g, wc) owns rows 64g.. x columns 64wc.. of the sub-tile (4 x 4 fragments).
// The accumulation order per output element is the one of every other sx_gemm kernel (residual as the initial value, k ascending,
// bias last): C and x16 are bit-identical to gemm_pp.hip's.
#include "gemm_common.h"
#include <type_traits>

namespace sxk_gemm {

#define ST_SYNC()                            \
  do {                                       \
    __builtin_amdgcn_sched_barrier(0);       \
    __builtin_amdgcn_s_barrier();            \
    __builtin_amdgcn_sched_barrier(0);       \
  } while (0)

template <int V>
using ic_t = std::integral_constant<int, V>;

// VM operations (besides its three DMAs) that barrier interval P of a sub-tile issues: fragment pieces 0..15 = fp32 store
// [+ the pair's 16-bit store on odd fragments] + residual load; 17 = the rows' statistics store
constexpr int strip_po(int P) { return P < 0 ? 0 : P < 16 ? ((P & 1) ? 3 : 2) : P == 17 ? 1 : 0; }
constexpr int kStripPeel = 10;   // k-tiles of a sub-tile whose intervals carry pieces (or follow one closely enough to change a wait count)

template <typename TT>
__global__ __launch_bounds__(512) void gemm_strip_kernel(const GemmP p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef typename TT::vec8 vec8;
  constexpr int BM = 128, BN = 256, A_BYTES = BM * 128, STAGE = A_BYTES + BN * 128, RING = 3 * STAGE;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* rsum = (float*)(smem + RING);             // [4 column waves][128 rows][2]
  float* bias_l = (float*)(smem + RING + 4096);    // [N]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2, wc = wave & 3;
  const int lq = (lane >> 4) * 4;
  const int nkt = p.K / 64;
  const int n_strips = p.M / BM;
  const int gstride = (int)gridDim.x * BM;
  const int my_strips = (n_strips - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int n_sub = my_strips * (p.N / BN);

  struct Sub { int row0, n0; };
  auto sub_next = [&](Sub s) -> Sub {
    s.n0 += BN;
    if (s.n0 >= p.N) { s.n0 = 0; s.row0 += gstride; }
    return s;
  };

  for (int i = tid; i < p.N; i += 512) bias_l[i] = p.bias ? p.bias[i] : 0.f;

  __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)p.a_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (int)p.w_bytes, 0x00020000);
  const int c_bytes = p.M * p.ldc * 4, x_bytes = p.M * p.ln_ldx * 2, s_bytes = p.M * 16;
  // descriptors of the previous sub-tile's outputs / the next sub-tile's residual: real, or NULL (0 records: stores dropped, loads 0)
  auto rsrc_c = [&](bool ok) { return __builtin_amdgcn_make_buffer_rsrc((void*)p.C, 0, ok ? c_bytes : 0, 0x00020000); };
  auto rsrc_x = [&](bool ok) { return __builtin_amdgcn_make_buffer_rsrc((void*)p.ln_x16, 0, ok ? x_bytes : 0, 0x00020000); };
  auto rsrc_r = [&](bool ok) { return __builtin_amdgcn_make_buffer_rsrc((void*)p.residual, 0, ok ? c_bytes : 0, 0x00020000); };
  auto rsrc_s = [&](bool ok) { return __builtin_amdgcn_make_buffer_rsrc((void*)p.ln_out, 0, ok ? s_bytes : 0, 0x00020000); };

  // ---- per-lane offsets (fixed over the whole launch): ONE VGPR per access kind; everything wave-uniform (slot, fragment, sub-tile,
  // k-tile) travels in the instructions' scalar offset, so nothing per-fragment is hoisted into registers -------------------------
  const int rl = lane >> 3;
  const int dma_lane = rl * p.K * 2 + (((lane & 7) ^ rl) << 4);          // row rl of an 8-row slot, swizzled 16-B chunk
  // C / residual (ldc == ldr, fp32): fragment (i, j) = rows 64g + 16j + (lane & 15), columns 64wc + 16i + lq .. +3
  const int c_lane = ((lane & 15) * p.ldc + lq) * 4;
  // x16: after the permlane swap of a fragment pair a lane owns 8 consecutive columns of the pair's 32.
  // (the rarely used lane constants are RE-COMPUTED where they are used, from a lane id hipcc cannot hoist: kept live across the
  // launch they are the registers that spill — and a scratch reload inside a piece sits on the same vmcnt as the counted waits)
  auto lane_now = [&]() -> int {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
  };
  auto x_lane_now = [&]() -> int {
    const int l = lane_now();
    return ((l & 15) * p.ln_ldx + ((l >> 4) & 1) * 16 + (l >> 5) * 8) * 2;
  };
  const int slot_pitch = 8 * p.K * 2;                                      // bytes between consecutive 8-row DMA slots
  const int c_wave = (g * 64 * p.ldc + wc * 64) * 4, c_j = 16 * p.ldc * 4;   // uniform parts of a fragment's C offset
  const int x_wave = (g * 64 * p.ln_ldx + wc * 64) * 2, x_j = 16 * p.ln_ldx * 2;
  const unsigned frag_row = (unsigned)(lane & 15) * 128u;
  unsigned frag_sw[2];
  frag_sw[0] = (unsigned)(((lane >> 4)) ^ (lane & 7)) << 4;
  frag_sw[1] = (unsigned)((4 + (lane >> 4)) ^ (lane & 7)) << 4;
  const unsigned a_frag = (unsigned)(g * 64) * 128u + frag_row;
  const unsigned w_frag = (unsigned)A_BYTES + (unsigned)(wc * 64) * 128u + frag_row;

  // ---- DMA stream: global k-tile counter over all sub-tiles of this workgroup (saturates on the last k-tile: harmless re-fetch) ----
  Sub isub = {(int)blockIdx.x * BM, 0};
  int ikt = 0, ileft = n_sub * nkt;
  int wr_stage = 0, rd_stage = 0;
  auto issue_half = [&](int h) {    // h = 0: A slots + W slot 0; h = 1: W slots 1..3   (3 DMA instructions each)
    unsigned char* sb = smem + wr_stage * STAGE;
    const int ua = isub.row0 * p.K * 2 + ikt * 128 + (2 * wave) * slot_pitch;
    const int uw = isub.n0 * p.K * 2 + ikt * 128 + (4 * wave) * slot_pitch;
    if (h == 0) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, SX_LDS_PTR(sb + (2 * wave) * 1024), 16, dma_lane, ua, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, SX_LDS_PTR(sb + (2 * wave + 1) * 1024), 16, dma_lane, ua + slot_pitch, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, SX_LDS_PTR(sb + A_BYTES + (4 * wave) * 1024), 16, dma_lane, uw, 0, 0);
    } else {
#pragma unroll
      for (int i = 1; i < 4; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, SX_LDS_PTR(sb + A_BYTES + (4 * wave + i) * 1024), 16, dma_lane, uw + i * slot_pitch, 0, 0);
    }
  };
  auto issue_advance = [&]() {
    wr_stage = wr_stage == 2 ? 0 : wr_stage + 1;
    if (ileft > 1) {
      --ileft;
      if (++ikt == nkt) { ikt = 0; isub = sub_next(isub); }
    }
  };

  f32x4_t acc0[4][4], acc1[4][4];     // [n-fragment i][m-fragment j]
  float lsum[4], lsq[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) lsum[j] = lsq[j] = 0.f;

  // sub-tile state: cur (accumulating), pv (being stored), nx (residual being loaded)
  Sub cur = {(int)blockIdx.x * BM, 0};
  Sub pv = cur, nx = cur;
  bool pv_ok = false, nx_ok = false, pv_last = false;
  __amdgpu_buffer_rsrc_t rCp = rsrc_c(false), rXp = rsrc_x(false), rRn = rsrc_r(false), rSp = rsrc_s(false);
  int pv_c = 0, pv_x = 0, pv_s = 0, nx_r = 0, pv_b = 0;
  auto set_neighbours = [&](int t) {
    nx = sub_next(cur);
    nx_ok = t + 1 < n_sub;
    pv_last = pv_ok && pv.n0 + BN >= p.N;
    rCp = rsrc_c(pv_ok); rXp = rsrc_x(pv_ok); rRn = rsrc_r(nx_ok);
    rSp = rsrc_s(pv_last && wc == 0);
    pv_c = (pv.row0 * p.ldc + pv.n0) * 4 + c_wave;
    pv_x = (pv.row0 * p.ln_ldx + pv.n0) * 2 + x_wave;
    pv_s = (pv.row0 + g * 64) * 16;
    pv_b = (pv.n0 + wc * 64) * 4;
    nx_r = (nx.row0 * p.ldc + nx.n0) * 4 + c_wave;
  };

  u32x2_t pk_keep = {0u, 0u};
  auto pack4 = [&](f32x4_t x) -> u32x2_t {
    u32x2_t o;
    o[0] = pack2<TT>(x[0], x[1]);
    o[1] = pack2<TT>(x[2], x[3]);
    return o;
  };
  // piece P of a sub-tile's pipeline, on accumulator set E (the one NOT accumulating)
  auto piece = [&](auto PC, f32x4_t (&accE)[4][4]) {
    constexpr int P = decltype(PC)::value;
    if constexpr (P >= 0 && P < 16) {
      constexpr int i = P & 3, j = P >> 2;
      const f32x4_t b = *(const f32x4_t*)((const unsigned char*)bias_l + (pv_b + i * 64) + (lane_now() >> 4) * 16);
      const f32x4_t v = accE[i][j] + b;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rCp, c_lane, pv_c + j * c_j + i * 64, 0);
      lsum[j] += (v[0] + v[1]) + (v[2] + v[3]);
      lsq[j] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
      asm volatile("" : "+v"(lsum[j]), "+v"(lsq[j]));     // pin: hipcc otherwise sinks all 16 fragments' sums to piece 16 (64 live registers)
      const u32x2_t pk = pack4(v);
      if constexpr ((i & 1) == 0) {
        pk_keep = pk;
      } else {
        const auto w0 = __builtin_amdgcn_permlane16_swap(pk_keep[0], pk[0], false, false);
        const auto w1 = __builtin_amdgcn_permlane16_swap(pk_keep[1], pk[1], false, false);
        const u32x4_t w4 = {w0[0], w1[0], w0[1], w1[1]};
        __builtin_amdgcn_raw_buffer_store_b128(w4, rXp, x_lane_now(), pv_x + j * x_j + (i - 1) * 32, 0);
      }
      accE[i][j] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rRn, c_lane, nx_r + j * c_j + i * 64, 0));
    } else if constexpr (P == 16) {
      // the strip's row sums (complete once its last sub-tile has gone through the pieces): four column quads of a wave → one
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float s1 = lsum[j], s2 = lsq[j];
        s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        // (after the two exchanges all four lanes of a row hold the same sums: every lane stores, no exec mask, no branch)
        *(f32x2_t*)(rsum + (size_t)(wc * BM + g * 64 + j * 16 + (lane_now() & 15)) * 2) = (f32x2_t){s1, s2};
        // a strip's sums restart behind its last sub-tile — and behind the pieces of the very first interval sequence, which ran on
        // an empty accumulator set with nothing to store (NULL descriptors) but did add its bias rows into the sums
        lsum[j] = (pv_last || !pv_ok) ? 0.f : lsum[j];
        lsq[j] = (pv_last || !pv_ok) ? 0.f : lsq[j];
      }
    } else if constexpr (P == 17) {
      // (a barrier later) the four column waves in a fixed order; one 16-byte store per row by the wc = 0 wave of each row group
      // (every wave issues the instruction: the others', and the one of a strip that is not complete yet, hit the NULL descriptor)
      const int ln = lane_now();
      const int r = g * 64 + ln;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const f32x2_t t = *(const f32x2_t*)(rsum + (size_t)(w * BM + r) * 2);
        s1 += t[0]; s2 += t[1];
      }
      typedef double f64x2_t __attribute__((ext_vector_type(2)));
      const f64x2_t d = {(double)s1, (double)s2};
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, d), rSp, ln * 16, pv_s, 0);
    }
  };

  vec8 af[4], wf[4];
  auto lgkm0 = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
  auto mma = [&](f32x4_t (&accM)[4][4]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // accumulator tied in place (with the builtin hipcc ping-pongs each accumulator between two register quads across the two
        // phases of the rolled loop: 128 registers for one set). No hazard inside: operands come from ds_reads behind lgkmcnt(0),
        // every accumulator is used once per segment, its VALU readers sit barrier intervals away.
        if constexpr (std::is_same<TT, F16>::value)
          asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(accM[i][j]) : "v"(wf[i]), "v"(af[j]));
        else
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(accM[i][j]) : "v"(wf[i]), "v"(af[j]));
      }
    __builtin_amdgcn_s_setprio(0);
  };
  auto read_frags = [&](int ks) {
    const unsigned char* sb = smem + rd_stage * STAGE;
#pragma unroll
    for (int j = 0; j < 4; ++j) af[j] = *(const vec8*)(sb + a_frag + j * 2048 + frag_sw[ks]);
#pragma unroll
    for (int i = 0; i < 4; ++i) wf[i] = *(const vec8*)(sb + w_frag + i * 2048 + frag_sw[ks]);
  };
  // one k-tile: intervals P0 = 2u, P1 = 2u + 1 of the sub-tile (or -1, -1 in the rolled part)
  auto ktile = [&](auto P0C, auto P1C, f32x4_t (&accM)[4][4], f32x4_t (&accE)[4][4]) {
    constexpr int P0 = decltype(P0C)::value, P1 = decltype(P1C)::value;
    constexpr int WAITN = 6 + strip_po(P0 - 1) + strip_po(P0) + strip_po(P1);
    // ---- k 0..31 ; first half of the DMA batch two k-tiles ahead ----
    issue_half(0);
    piece(P0C, accE);
    __builtin_amdgcn_sched_barrier(0);   // the piece's temporaries die before the 32 fragment registers are written (peak pressure)
    read_frags(0);
    lgkm0();
    ST_SYNC();
    mma(accM);
    ST_SYNC();
    // ---- k 32..63 ; second half ; the batch ONE k-tile ahead has landed (this wave's share) ----
    issue_half(1);
    issue_advance();
    piece(P1C, accE);
    __builtin_amdgcn_sched_barrier(0);
    read_frags(1);
    wait_vmcnt<WAITN>();
    lgkm0();
    ST_SYNC();
    mma(accM);
    ST_SYNC();
    rd_stage = rd_stage == 2 ? 0 : rd_stage + 1;
  };
  auto subtile = [&](int t, f32x4_t (&accM)[4][4], f32x4_t (&accE)[4][4]) {
    set_neighbours(t);
    ktile(ic_t<0>{}, ic_t<1>{}, accM, accE);
    ktile(ic_t<2>{}, ic_t<3>{}, accM, accE);
    ktile(ic_t<4>{}, ic_t<5>{}, accM, accE);
    ktile(ic_t<6>{}, ic_t<7>{}, accM, accE);
    ktile(ic_t<8>{}, ic_t<9>{}, accM, accE);
    ktile(ic_t<10>{}, ic_t<11>{}, accM, accE);
    ktile(ic_t<12>{}, ic_t<13>{}, accM, accE);
    ktile(ic_t<14>{}, ic_t<15>{}, accM, accE);
    ktile(ic_t<16>{}, ic_t<17>{}, accM, accE);
    ktile(ic_t<18>{}, ic_t<19>{}, accM, accE);
    for (int u = kStripPeel; u < nkt; ++u) ktile(ic_t<-1>{}, ic_t<-1>{}, accM, accE);
    pv = cur; pv_ok = true;
    cur = nx;
  };

  // ---- head: residual of sub-tile 0, the first two DMA batches ----------------------------------------------------------------
  {
    __amdgpu_buffer_rsrc_t rR0 = rsrc_r(true);
    const int r0 = (cur.row0 * p.ldc + cur.n0) * 4 + c_wave;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc0[i][j] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rR0, c_lane, r0 + j * c_j + i * 64, 0));
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc1[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }
  issue_half(0); issue_half(1); issue_advance();
  issue_half(0); issue_half(1); issue_advance();
  wait_vmcnt<0>();
  __syncthreads();                       // bias table + first two k-tiles
  if (g == 1) ST_SYNC();                 // group 1 runs one barrier interval behind group 0

  // the last sub-tile's outputs: nothing left to overlap them with. One copy per accumulator set (per loop exit): merging the two
  // exits in front of a shared tail makes hipcc shuffle / spill whole accumulator sets at the join
  auto finish = [&](f32x4_t (&accL)[4][4]) {
    if (g == 0) ST_SYNC();                 // pairs with group 1's last barrier
    wait_vmcnt<0>();                       // the saturated tail DMAs must not outlive the LDS allocation
    // pv / pv_ok were set by the last subtile(); cur is past the end
    nx = cur;
    nx_ok = false;
    pv_last = pv.n0 + BN >= p.N;
    rCp = rsrc_c(true); rXp = rsrc_x(true); rRn = rsrc_r(false);
    rSp = rsrc_s(pv_last && wc == 0);
    pv_c = (pv.row0 * p.ldc + pv.n0) * 4 + c_wave;
    pv_x = (pv.row0 * p.ln_ldx + pv.n0) * 2 + x_wave;
    pv_s = (pv.row0 + g * 64) * 16;
    pv_b = (pv.n0 + wc * 64) * 4;
    nx_r = 0;
    piece(ic_t<0>{}, accL); piece(ic_t<1>{}, accL); piece(ic_t<2>{}, accL); piece(ic_t<3>{}, accL);
    piece(ic_t<4>{}, accL); piece(ic_t<5>{}, accL); piece(ic_t<6>{}, accL); piece(ic_t<7>{}, accL);
    piece(ic_t<8>{}, accL); piece(ic_t<9>{}, accL); piece(ic_t<10>{}, accL); piece(ic_t<11>{}, accL);
    piece(ic_t<12>{}, accL); piece(ic_t<13>{}, accL); piece(ic_t<14>{}, accL); piece(ic_t<15>{}, accL);
    piece(ic_t<16>{}, accL);
    __syncthreads();
    piece(ic_t<17>{}, accL);
  };
  int t = 0;
  while (true) {
    subtile(t, acc0, acc1);
    if (++t >= n_sub) { finish(acc0); break; }
    subtile(t, acc1, acc0);
    if (++t >= n_sub) { finish(acc1); break; }
  }
#endif
}

bool strip_supported(const GemmP& p, int a_mode) {
  if (a_mode != SX_A_LINEAR || p.out_dtype != SX_F32 || !p.res_init || p.glu || p.act != SX_ACT_NONE) return false;
  if (!p.ln_out || !p.ln_x16 || p.ln_in || p.gn_stats || p.bias2d || p.res_mod) return false;
  if (p.K != p.Kw || p.K % 64 || p.K / 64 < kStripPeel) return false;
  if (p.N % 256 || p.N > 2560 || p.M % 128 || p.n_valid != p.N) return false;
  if (p.ldc != p.ldr || p.ldc % 4 || p.ln_ldx % 8 || (((size_t)p.ln_x16) & 15) || (((size_t)p.C) & 15) || (((size_t)p.residual) & 15)) return false;
  if ((uint64_t)p.M * p.ldc * 4 >= 0x7fffffffull || (uint64_t)p.M * p.ln_ldx * 2 >= 0x7fffffffull) return false;
  return true;
}

template <typename TT>
static int launch_strip_t(const GemmP& p, int cus, hipStream_t st) {
  const int strips = p.M / 128;
  const int grid = strips < cus ? strips : cus;
  const size_t lds = 3 * (size_t)(128 + 256) * 128 + 4096 + (size_t)p.N * 4;
  auto k = gemm_strip_kernel<TT>;
  static hipError_t attr[16];
  static bool done[16];
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev &= 15;
  if (!done[dev]) {
    attr[dev] = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (128 + 256) * 128 + 4096 + 2560 * 4);
    done[dev] = true;
  }
  SX_CHECK(attr[dev] == hipSuccess, "sx_gemm: cannot reserve LDS for the strip kernel: %s", hipGetErrorString(attr[dev]));
  hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, st, p);
  SX_HIP_LAUNCH_CHECK();
  return SX_OK;
}

int launch_strip(const GemmP& p, int dtype, hipStream_t st) {
  static int cus[16];
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev &= 15;
  if (!cus[dev]) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return SX_ERR_HIP;
    cus[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return dtype == SX_BF16 ? launch_strip_t<BF16>(p, cus[dev], st) : launch_strip_t<F16>(p, cus[dev], st);
}

}  // namespace sxk_gemm
