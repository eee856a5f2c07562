// Persistent "strip" GEMM for the fp32-residual / LayerNorm-producer launches of the UNet's transformer blocks (gfx950, MI355X):
//
//     C[M][N] (fp32) = residual[M][N] (fp32) + A[M][K] · W[N][K]^T + bias[N],     x16[M][N] = round16(C),   rows[M] = (Σ C, Σ C²)
//
// (attention out-projections and ff.net.2 of diffusers' BasicTransformerBlock [ext] as the t2i loop calls them,
// pipeline_stable_diffusion_xl_t2i_edit.py:915-922; same contract as gemm_pp.hip's LN = 2 producer epilogue, include/seedx_hip.h
// sx_gemm_ln). STATUS (round 6): correct — C / x16 bit-identical to the ping-pong producer, row sums reproducible bit for bit — and
// SLOWER than it on every production shape (217-264 vs 154 us on the 32768 x 1280 x 1280 out-projection): opt-in through
// sx_gemm_force_tile(9), never the automatic choice. The measurements and the reason (the CU's LDS-DMA issue path bounds a tile
// that needs twice the operand bytes per flop) are in profiles/r6_ab_experiments.md §1; the kernel stays in the tree as the
// measured answer to "overlap the producer's three phases inside one workgroup" and for its lab / probe build.
//
// These launches move 12 B per output element for 2·K flops: at K = 1280 their roofline is HBM, and the one-tile-per-workgroup
// kernels run three serial phases per tile on a CU that holds ONE workgroup — residual pre-load (HBM), main loop (MFMA, HBM idle),
// stores (HBM). Here the three overlap:
//
//   * one workgroup per CU walks whole 128-row STRIPS of the output (grid = min(strips, CUs)), 256 columns (a "sub-tile") at a time;
//   * TWO accumulator sets of 64 registers: while sub-tile t accumulates into set t & 1, the other set is — fragment by fragment, one
//     16x16 fragment per barrier interval of the first 16 phases — turned into sub-tile t-1's output (bias add, fp32 store, 16-bit
//     copy, row sums) and immediately re-loaded with sub-tile t+1's residual. Loads, stores and the operand LDS-DMA share vmcnt in
//     issue order, every access is a buffer instruction whose descriptor is the NULL descriptor when the neighbour sub-tile does not
//     exist, so each barrier interval issues a compile-time number of VM operations and the counted waits stay exact;
//   * the operand ring (6 stages of one 32-deep k-step: (128 A rows + 256 W rows) x 64 B) never drains: the DMA batch five phases
//     ahead is issued into the stage both wave groups finished reading one interval ago, across sub-tile and strip boundaries;
//   * a strip covers all N columns, so the rows' (Σ, Σ²) are complete inside one workgroup: lane partials → two xor shuffles → the
//     four column waves meet in LDS in a fixed order → ONE plain 16-byte store per row. No atomics.
//
// Schedule inside a phase: the 8-wave ping-pong of gemm_pp.hip (two groups of four waves one barrier interval apart; a group's load
// segment {8 fragment ds_reads, 1 DMA issue, its epilogue piece, counted vmcnt, lgkmcnt(0)} runs under the other group's 16 MFMAs,
// which carry the phase's other two DMA issues in their issue gaps); wave (g, wc) owns rows 64g.. x columns 64wc.. of the sub-tile
// (4 x 4 fragments). The accumulation order per output element is the one of every other sx_gemm kernel (residual as the initial
// value, k ascending, bias last): C and x16 are bit-identical to gemm_pp.hip's.
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
constexpr int kStripPeel = 21;   // phases of a sub-tile that carry pieces (0..17) or follow one closely enough to change a wait count

// PROBE: tuning build (sx_gemm_debug_stamps): every wave sums the s_memtime spans of the six parts of its ROLLED phases and stores
// them at the end — dbg[(block * 8 + wave) * 8 + {0: issue (fragment reads + DMA), 1: counted vmcnt, 2: lgkmcnt(0), 3: first barrier,
// 4: MFMA segment, 5: second barrier, 6: phases counted, 7: whole kernel}]
template <typename TT, int ABL = -1>     // ABL >= 0: probe build with compile-time ablation mask (1 = no W fragment reads, 2 = no DMA, 4 = no A fragment reads, 8 = no MFMA: timing only)
__global__ __launch_bounds__(512) void gemm_strip_kernel(const GemmP p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef typename TT::vec8 vec8;
  constexpr bool PROBE = ABL >= 0;
  // operand ring: NSTG stages of ONE 32-deep k-step each = (128 A rows + 256 W rows) x 64 B = 24 KB. A CU that runs 128x256 sub-tiles
  // consumes 48 KB of operands per 64-deep k-tile in ~0.55 us of MFMA time — twice the bytes per flop of a 256x320 tile — so what bounds
  // it is (bytes in flight) / (memory latency): with three 64-deep stages a DMA had 0.5-1.5 k-tiles between issue and wait and the
  // loop ran at 1.1-1.5 us per k-tile (latency-bound even on an idle chip); with six 32-deep stages every stage is re-issued one
  // barrier interval after its last read and has four phases (two k-tiles) to land
  constexpr int BM = 128, BN = 256, NSTG = 6, A_BYTES = BM * 64, STAGE = A_BYTES + BN * 64, RING = NSTG * STAGE;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* rsum = (float*)(smem + RING);             // [4 column waves][128 rows][2]
  float* bias_l = (float*)(smem + RING + 4096);    // [N]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2, wc = wave & 3;
  const int lq = (lane >> 4) * 4;
  const int nph = p.K / 32;                        // 32-deep phases per sub-tile
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
  // LDS image of a stage: 1-KB blocks of 16 rows x 64 B (= one MFMA operand fragment = one DMA instruction). Chunk c (16 B) of row r
  // sits at r * 64 + ((c ^ sw(r >> 2)) << 4), sw = {0, 2, 3, 1}: the four 16-lane groups a ds_read_b128 is served in then hit 16
  // different 16-B bank groups each (rows r, r + 4, r + 8, r + 12 share a 256-B bank row)
  auto sw4 = [](int r) -> int { return (0x78 >> (2 * ((r >> 2) & 3))) & 3; };       // r >> 2 = 0, 1, 2, 3 -> 0, 2, 3, 1 (2-bit table)
  const int dma_lane = (lane >> 2) * p.K * 2 + (((lane & 3) ^ sw4(lane >> 2)) << 4);   // row lane >> 2 of a 16-row slot, the chunk that
                                                                                       // belongs at physical position lane & 3
  // C / residual (ldc == ldr, fp32): fragment (i, j) = rows 64g + 16j + (lane & 15), columns 64wc + 16i + lq .. +3
  const int c_lane = ((lane & 15) * p.ldc + lq) * 4;       // (head only; the pieces re-compute it: c_lane_now)
  // x16: after the permlane swap of a fragment pair a lane owns 8 consecutive columns of the pair's 32.
  // (the rarely used lane constants are RE-COMPUTED where they are used, from a lane id hipcc cannot hoist: kept live across the
  // launch they are the registers that spill — and a scratch reload inside a piece sits on the same vmcnt as the counted waits)
  auto lane_now = [&]() -> int {
    int l;
    // s_nop 1: hipcc pads no hazard of an asm statement — its output register may be the DATA register of the 16-byte buffer_store
    // issued just before (a VALU write needs two wait states behind such a store; without them the stores that sat in a backed-up
    // VMEM queue wrote lane numbers instead of x16 values: 1 row in 10^5 under load, none on an idle chip)
    asm volatile("s_nop 1\n\tv_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
  };
  auto c_lane_now = [&]() -> int {
    const int l = lane_now();
    return ((l & 15) * p.ldc + (l >> 4) * 4) * 4;
  };
  auto x_lane_now = [&]() -> int {
    const int l = lane_now();
    return ((l & 15) * p.ln_ldx + ((l >> 4) & 1) * 16 + (l >> 5) * 8) * 2;
  };
  const int slot_pitch = 16 * p.K * 2;                                     // bytes between consecutive 16-row DMA slots
  const int c_wave = (g * 64 * p.ldc + wc * 64) * 4, c_j = 16 * p.ldc * 4;   // uniform parts of a fragment's C offset
  const int x_wave = (g * 64 * p.ln_ldx + wc * 64) * 2, x_j = 16 * p.ln_ldx * 2;
  // fragment reads: ONE per-lane register (row lane & 15, chunk lane >> 4 of a 1-KB block); A fragment j of row group g is block
  // 4g + j of the stage, W fragment i of column wave wc block 8 + 4wc + i (wave-uniform distances)
  const unsigned frag0 = (unsigned)(lane & 15) * 64u + ((unsigned)((lane >> 4) ^ sw4(lane & 15)) << 4);
  const unsigned a_blk = (unsigned)(g * 4) * 1024u, w_blk = (unsigned)A_BYTES + (unsigned)(wc * 4) * 1024u;

  // ---- DMA stream: global phase counter over all sub-tiles of this workgroup (saturates on the last phase: harmless re-fetch) ----
  Sub isub = {(int)blockIdx.x * BM, 0};
  int iph = 0, ileft = n_sub * nph;
  int wr_stage = 0, rd_stage = 0;
  // DMA slot d of the batch the stream points at: 0 = the wave's A slot (rows 16 wave ..), 1, 2 = its two W slots
  auto issue_slot = [&](int d) {
    if constexpr ((PROBE ? ABL : 0) & 2) return;
    unsigned char* sb = smem + wr_stage * STAGE;
    if (d == 0) {
      const int ua = isub.row0 * p.K * 2 + iph * 64 + wave * slot_pitch;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, SX_LDS_PTR(sb + wave * 1024), 16, dma_lane, ua, 0, 0);
    } else {
      const int uw = isub.n0 * p.K * 2 + iph * 64 + (2 * wave + d - 1) * slot_pitch;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, SX_LDS_PTR(sb + A_BYTES + (2 * wave + d - 1) * 1024), 16, dma_lane, uw, 0, 0);
    }
  };
  auto issue_advance = [&]() {
    wr_stage = wr_stage == NSTG - 1 ? 0 : wr_stage + 1;
    if (ileft > 1) {
      --ileft;
      if (++iph == nph) { iph = 0; isub = sub_next(isub); }
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
      const int cl = c_lane_now();
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rCp, cl, pv_c + j * c_j + i * 64, 0);
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
      accE[i][j] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rRn, cl, nx_r + j * c_j + i * 64, 0));
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

  unsigned long long prb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long prb_start = PROBE ? __builtin_amdgcn_s_memtime() : 0ull;
  vec8 af[4], wf[4];
  auto lgkm0 = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
  // MFMA segment: 16 MFMAs, accumulators tied in place (with the builtin hipcc ping-pongs each accumulator between two register
  // quads across the two phases of the rolled loop: 128 registers for one set; no hazard inside: operands come from ds_reads behind
  // lgkmcnt(0), every accumulator is used once per segment, its VALU readers sit barrier intervals away). Two of the phase's three
  // DMA issues ride in the issue gaps of the MFMA stream (a 16x16x32 MFMA occupies the pipe for 16 cycles, its issue takes 4): the
  // load segment keeps 8 fragment reads + 1 DMA + its piece
  auto mma = [&](f32x4_t (&accM)[4][4]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr ((PROBE ? ABL : 0) & 8) continue;
        if constexpr (std::is_same<TT, F16>::value)
          asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(accM[i][j]) : "v"(wf[i]), "v"(af[j]));
        else
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(accM[i][j]) : "v"(wf[i]), "v"(af[j]));
      }
      if (i == 0 || i == 2) {
        __builtin_amdgcn_sched_barrier(0);
        issue_slot(1 + (i >> 1));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };
  constexpr int abl = PROBE ? ABL : 0;
  auto read_frags = [&]() {
    const unsigned char* sa = smem + (frag0 + (unsigned)(rd_stage * STAGE)) + a_blk;
    const unsigned char* sw = smem + (frag0 + (unsigned)(rd_stage * STAGE)) + w_blk;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if constexpr (!(abl & 4)) af[j] = *(const vec8*)(sa + j * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if constexpr (!(abl & 1)) wf[i] = *(const vec8*)(sw + i * 1024);
  };
  // one 32-deep phase (interval P of the sub-tile, or -1 in the rolled part). The stream issues the batch FIVE phases ahead (into the
  // stage both groups finished reading one interval ago); the counted wait retires the batch ONE phase ahead (this wave's share):
  // issued four phases back, visible to every wave once all of them have passed this wait and the barrier behind it.
  auto phase = [&](auto PC, f32x4_t (&accM)[4][4], f32x4_t (&accE)[4][4]) {
    constexpr int P = decltype(PC)::value;
    // VM operations issued behind the LAST DMA (slot 2, in an MFMA segment) of the batch issued four phases back
    constexpr int WAITN = 10 + strip_po(P - 3) + strip_po(P - 2) + strip_po(P - 1) + strip_po(P);
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0;
    if constexpr (PROBE && P < 0) t0 = __builtin_amdgcn_s_memtime();
    issue_advance();
    read_frags();
    issue_slot(0);
    piece(PC, accE);
    if constexpr (PROBE && P < 0) t1 = __builtin_amdgcn_s_memtime();
    wait_vmcnt<WAITN>();
    if constexpr (PROBE && P < 0) t2 = __builtin_amdgcn_s_memtime();
    lgkm0();
    if constexpr (PROBE && P < 0) t3 = __builtin_amdgcn_s_memtime();
    ST_SYNC();
    if constexpr (PROBE && P < 0) t4 = __builtin_amdgcn_s_memtime();
    mma(accM);
    if constexpr (PROBE && P < 0) t5 = __builtin_amdgcn_s_memtime();
    ST_SYNC();
    if constexpr (PROBE && P < 0) {
      const unsigned long long t6 = __builtin_amdgcn_s_memtime();
      prb[0] += t1 - t0; prb[1] += t2 - t1; prb[2] += t3 - t2; prb[3] += t4 - t3; prb[4] += t5 - t4; prb[5] += t6 - t5; prb[6] += 1;
    }
    rd_stage = rd_stage == NSTG - 1 ? 0 : rd_stage + 1;
  };
  auto subtile = [&](int t, f32x4_t (&accM)[4][4], f32x4_t (&accE)[4][4]) {
    set_neighbours(t);
    phase(ic_t<0>{}, accM, accE); phase(ic_t<1>{}, accM, accE); phase(ic_t<2>{}, accM, accE); phase(ic_t<3>{}, accM, accE);
    phase(ic_t<4>{}, accM, accE); phase(ic_t<5>{}, accM, accE); phase(ic_t<6>{}, accM, accE); phase(ic_t<7>{}, accM, accE);
    phase(ic_t<8>{}, accM, accE); phase(ic_t<9>{}, accM, accE); phase(ic_t<10>{}, accM, accE); phase(ic_t<11>{}, accM, accE);
    phase(ic_t<12>{}, accM, accE); phase(ic_t<13>{}, accM, accE); phase(ic_t<14>{}, accM, accE); phase(ic_t<15>{}, accM, accE);
    phase(ic_t<16>{}, accM, accE); phase(ic_t<17>{}, accM, accE); phase(ic_t<18>{}, accM, accE); phase(ic_t<19>{}, accM, accE);
    phase(ic_t<20>{}, accM, accE);
    for (int u = kStripPeel; u < nph; ++u) phase(ic_t<-1>{}, accM, accE);
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
  for (int b = 0; b < NSTG - 1; ++b) {    // the first five phases' operands
    if (b) issue_advance();
    for (int d = 0; d < 3; ++d) issue_slot(d);
  }
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
    if constexpr (PROBE) {
      prb[7] = __builtin_amdgcn_s_memtime() - prb_start;
      if (p.dbg && lane == 0)
        for (int q = 0; q < 8; ++q) p.dbg[((size_t)blockIdx.x * 8 + wave) * 8 + q] = prb[q];
    }
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
  if (p.K != p.Kw || p.K % 64 || p.K / 32 < kStripPeel) return false;
  if (p.N % 256 || p.N > 2560 || p.M % 128 || p.n_valid != p.N) return false;
  if (p.ldc != p.ldr || p.ldc % 4 || p.ln_ldx % 8 || (((size_t)p.ln_x16) & 15) || (((size_t)p.C) & 15) || (((size_t)p.residual) & 15)) return false;
  if ((uint64_t)p.M * p.ldc * 4 >= 0x7fffffffull || (uint64_t)p.M * p.ln_ldx * 2 >= 0x7fffffffull) return false;
  return true;
}

template <typename TT>
static int launch_strip_t(const GemmP& p0, int cus, hipStream_t st) {
  GemmP p = p0;
  p.dbg = g_dbg;
  // (probe builds: sx_gemm_debug_stamps + sx_gemm_force_tile(300 + ablation mask))
  const int strips = p.M / 128;
  const int grid = strips < cus ? strips : cus;
  const size_t lds = 6 * (size_t)(128 + 256) * 64 + 4096 + (size_t)p.N * 4;
  void (*k)(const GemmP) = gemm_strip_kernel<TT, -1>;
  if (g_dbg) {
    k = gemm_strip_kernel<TT, 0>;
#if defined(SX_STRIP_ABLATIONS)   // tools/lab/gemm_lab stripprobe: compile-time ablation masks (profiles/r6_strip_ablations.log)
    switch (g_gm) {
      case 1: k = gemm_strip_kernel<TT, 1>; break;
      case 2: k = gemm_strip_kernel<TT, 2>; break;
      case 4: k = gemm_strip_kernel<TT, 4>; break;
      case 5: k = gemm_strip_kernel<TT, 5>; break;
      case 8: k = gemm_strip_kernel<TT, 8>; break;
      default: break;
    }
#endif
  }
  static hipError_t attr[16];
  static bool done[16];
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev &= 15;
  if (!done[dev]) {
    const int ldsmax = 6 * (128 + 256) * 64 + 4096 + 2560 * 4;
    attr[dev] = hipFuncSetAttribute((const void*)gemm_strip_kernel<TT, -1>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsmax);
#if defined(SX_STRIP_ABLATIONS)
    const void* probes[6] = {(const void*)gemm_strip_kernel<TT, 0>, (const void*)gemm_strip_kernel<TT, 1>, (const void*)gemm_strip_kernel<TT, 2>,
                             (const void*)gemm_strip_kernel<TT, 4>, (const void*)gemm_strip_kernel<TT, 5>, (const void*)gemm_strip_kernel<TT, 8>};
#else
    const void* probes[1] = {(const void*)gemm_strip_kernel<TT, 0>};
#endif
    for (const void* q : probes)
      if (attr[dev] == hipSuccess) attr[dev] = hipFuncSetAttribute(q, hipFuncAttributeMaxDynamicSharedMemorySize, ldsmax);
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
