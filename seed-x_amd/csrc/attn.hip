// Fused attention for gfx950 (MI355X): flash-style MFMA kernel + small generic VALU kernel.
//
// MFMA kernel:
//   * block = 4 waves, 128 query rows (32 per wave); KV tile = 64 keys; 2-stage LDS ring filled by
//     `buffer_load_dwordx4 … lds` DMA (K tile [64][DP], V tile [64][DP], both row-major); one barrier per KV tile
//   * S^T = K · Q^T with v_mfma_f32_32x32x16: the lane that owns query q = lane&31 holds 16 of the 32
//     scores of a key block in registers → row max / row sum are lane-local plus ONE lane^32 exchange
//   * O^T = V^T · P^T reuses the score registers directly as the MFMA B operand: the contraction order
//     over keys is permuted by loading the K rows of a tile in the order r ↔ key swap_bits23(r): lane-half hi then owns
//     the keys 8hi..8hi+7 of every 16-key step = one 16-B piece of a V^T row (single ds_read_b128), so
//     no cross-lane shuffle / LDS round trip of P is needed
//   * V stays in its natural row-major [key][d] layout in HBM and in LDS (DMA'd exactly like K): the key-contiguous V^T
//     fragment the P·V MFMA wants is produced by gfx950's transposing LDS read `ds_read_b64_tr_b16` (two per 8-key
//     fragment) — no V^T pre-pass over HBM. Routing of that instruction (probed on the MI355X, tools/probes/
//     tr_read_probe.hip): in each 16-lane group, lane l receives element (l & 3) of the 8-byte rows addressed by lanes
//     4j + (l >> 2), j = 0..3. The V tile's 16-B chunks are XOR-swizzled by row so the 4 rows x 2 groups of one
//     32-lane LDS cycle fall on 64 distinct banks
//   * head_dim 104 (ViT-G, qwen_visual.py:170) is zero-padded to 128 only in LDS/registers via the buffer
//     range check — HBM layout stays [.., 104]
#include "sx_common.h"
#include <type_traits>

namespace sxk_attn {


template <int N>
__device__ __forceinline__ void wait_vmcnt_attn() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

typedef short tr4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) tr4_t lds_tr4_t;

struct AttnP {
  const unsigned short* Q;
  const unsigned short* K;
  const unsigned short* V;
  unsigned short* O;
  int B, H, Sq, Skv, D, q_tiles, causal;
  long long qbs, qrs, qhs, kbs, krs, khs, vbs, vrs, vhs, obs, ors;
  float scale_log2;
};

// OPT (bit mask, A/B-able through sx_attention_variant(16 + OPT); 0 = round-3 kernel):
//   1  s_setprio 1 around the two MFMA clusters of a tile: with 3-4 workgroups per CU every SIMD holds waves of different workgroups in
//      different phases; the arbiter is age-first, so an older wave grinding through its ~130 softmax VALU issues starves a younger
//      wave's MFMA cluster and the matrix pipe idles (MI355X_MICROARCH.md §Two waves per SIMD item 2; cdna_hip_programming.md T5)
//   2  lane^32 exchange of the row maximum by v_permlane32_swap (VALU) instead of ds_bpermute (LDS queue, behind the fragment reads)
//   4  row sums from the matrix pipe: one extra 32x32x16 MFMA per 16 keys with an all-ones A operand (every output row = the sum
//      over all 16 keys of both lane halves) instead of 16 v_dot2 per tile; the sum is over the same ROUNDED probabilities
//   8  software pipeline over KV tiles (cdna_hip_programming.md T15): S(t+1) = K(t+1)·Qᵀ is issued BEFORE the softmax of tile t, so
//      one wave's instruction stream carries independent MFMAs next to its VALU-bound softmax; K tiles are staged one tile ahead
//      of V tiles (same two 16-KB buffers). Costs 32 more live VGPRs → 3 workgroups per CU instead of 4 (measured neutral in round 3)
//  16  (with 8) the S(t+1) MFMAs sit INSIDE the softmax's basic block, spread between its VALU instructions by
//      sched_group_barrier (one MFMA per ~12 VALU / transcendental issues): an in-order wave only overlaps its own matrix and vector
//      work when the two are interleaved in program order; as a leading cluster (8 alone) the 8 MFMAs stall issue for 8 x 32 cycles
template <typename TT, int DP, int OPT = 0>
__global__ __launch_bounds__(256, (DP <= 64 ? ((OPT & 24) ? 3 : 4) : 1)) void attn_kernel(const AttnP p) {  // D=64: 4 waves per SIMD (<= 128 VGPRs)
#if defined(__HIP_DEVICE_COMPILE__)  // body uses gfx950-only builtins (LDS-DMA, MFMA); the host pass only needs the stub
  typedef typename TT::vec8 vec8;
  typedef typename TT::vec4 vec4;
  constexpr bool PRIO = (OPT & 1) != 0, SWAP = (OPT & 2) != 0, ONES = (OPT & 4) != 0, PIPE = (OPT & 24) != 0, ILV = (OPT & 16) != 0;
  constexpr bool DEFER = (OPT & 32) != 0;
  constexpr int KROW = DP * 2;                 // bytes per K row in LDS (128 | 256)
  constexpr int K_BYTES = 64 * KROW;           // K tile
  constexpr int V_BYTES = K_BYTES;             // V tile: 64 keys x DP, row-major like K
  constexpr int STAGE = K_BYTES + V_BYTES;
  constexpr int K_SLOTS = K_BYTES / 1024, V_SLOTS = V_BYTES / 1024;  // 1-KiB DMA slots
  constexpr int KCH = KROW / 16;               // 16-B chunks per K row (8 | 16)
  constexpr int KRPS = 1024 / KROW;            // K rows per slot (8 | 4)
  constexpr int NKS = DP / 16;                 // k-steps of the S^T MFMA
  constexpr int NDT = DP / 32;                 // 32-wide d tiles of O^T
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, lq = lane & 31;

  const int t = xcd_remap(blockIdx.x, gridDim.x);
  const int qt = t % p.q_tiles, bh = t / p.q_tiles;
  const int b = bh / p.H, h = bh % p.H;
  const int q0 = qt * 128;
  const int coff = p.Skv - p.Sq;  // causal: key j visible iff j <= q + coff

  int nt = (p.Skv + 63) / 64;
  if (p.causal) {
    const int last = q0 + 127 + coff;  // largest visible key of this q block
    const int ntc = last < 0 ? 0 : last / 64 + 1;
    nt = ntc < nt ? ntc : nt;
  }

  const unsigned short* Kb = p.K + b * p.kbs + h * p.khs;
  const unsigned short* Vb = p.V + b * p.vbs + h * p.vhs;
  const int k_bytes = (int)(((long long)(p.Skv - 1) * p.krs + p.D) * 2);
  const int v_bytes = (int)(((long long)(p.Skv - 1) * p.vrs + p.D) * 2);
  __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc((void*)Kb, 0, k_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc((void*)Vb, 0, v_bytes, 0x00020000);

  // ---- per-lane DMA sources -------------------------------------------------------------------
  // K: slot s holds KRPS rows; lane -> (row_in_slot, chunk position); logical chunk = pos ^ key(row)
  unsigned k_off[K_SLOTS / 4];
  int k_row[K_SLOTS / 4];
#pragma unroll
  for (int i = 0; i < K_SLOTS / 4; ++i) {
    const int slot = wave * (K_SLOTS / 4) + i;
    const int r = slot * KRPS + lane / KCH;  // row within the 64-key tile
    const int pos = lane % KCH;
    const int key = (KCH == 16) ? (r & 15) : ((r >> 1) & 7);
    const int c = pos ^ key;
    // LDS row r holds KEY perm(r) = r with bits 2 and 3 swapped: the 32x32 MFMA leaves lane-half `hi` with S^T rows
    // {4hi+i, 8+4hi+i, ...}; with this order those are the keys 8hi .. 8hi+7 of each 16-key step, i.e. ONE contiguous
    // 16-B piece of a V^T row, so the P·V operand is a single ds_read_b128 (no 8-B halves to stitch with v_mov)
    k_row[i] = (r & ~0xC) | ((r & 4) << 1) | ((r & 8) >> 1);
    k_off[i] = (c * 8 < p.D) ? (unsigned)(c * 16) : 0x80000000u;
  }
  // V: slot s holds KRPS rows (keys, natural order) of KROW bytes; lane -> (row_in_slot, chunk position); the logical
  // 16-B chunk stored at position pos of row r is pos ^ vswz(r), vswz chosen for the transposing reads (see header)
  unsigned v_off[V_SLOTS / 4];
  int v_row[V_SLOTS / 4];
#pragma unroll
  for (int i = 0; i < V_SLOTS / 4; ++i) {
    const int slot = wave * (V_SLOTS / 4) + i;
    const int r = slot * KRPS + lane / KCH;
    const int pos = lane % KCH;
    const int key = (KCH == 16) ? ((r & 3) << 2) : (((r >> 1) & 1) << 2);
    const int c = pos ^ key;
    v_row[i] = r;
    v_off[i] = (c * 8 < p.D) ? (unsigned)(c * 16) : 0x80000000u;
  }

  auto stage_k = [&](int buf, int kvt) {
    unsigned char* sK = smem + buf * STAGE;
    const int kv0 = kvt * 64;
#pragma unroll
    for (int i = 0; i < K_SLOTS / 4; ++i) {
      const int kv = kv0 + k_row[i];
      const unsigned voff = (kv < p.Skv) ? (unsigned)((long long)kv * p.krs * 2) + k_off[i] : 0x80000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rK, SX_LDS_PTR(sK + (wave * (K_SLOTS / 4) + i) * 1024), 16, voff, 0,
                                               0, 0);
    }
  };
  auto stage_v = [&](int buf, int kvt) {
    unsigned char* sV = smem + buf * STAGE + K_BYTES;
    const int kv0 = kvt * 64;
#pragma unroll
    for (int i = 0; i < V_SLOTS / 4; ++i) {
      const int kv = kv0 + v_row[i];                 // rows past Skv are fetched as zeros (0 · P, never NaN)
      const unsigned voff = (kv < p.Skv) ? (unsigned)((long long)kv * p.vrs * 2) + v_off[i] : 0x80000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rV, SX_LDS_PTR(sV + (wave * (V_SLOTS / 4) + i) * 1024), 16, voff, 0, 0,
                                               0);
    }
  };

  // ---- Q fragments (B operand of S^T = K·Q^T): lane (q = lane&31, hi) holds Q[q][16ks + 8hi .. +7] ------
  const int qrow = q0 + wave * 32 + lq;
  vec8 qf[NKS];
  {
    const unsigned short* Qr = p.Q + b * p.qbs + (long long)qrow * p.qrs + h * p.qhs;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int d = ks * 16 + hi * 8;
      u32x4_t raw = {0u, 0u, 0u, 0u};
      if (qrow < p.Sq && d < p.D) raw = *(const u32x4_t*)(Qr + d);
      __builtin_memcpy(&qf[ks], &raw, 16);
    }
  }

  f32x16_t o[NDT];
#pragma unroll
  for (int i = 0; i < NDT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f, mc_run = 0.f;

  // LDS read offsets
  // K frag (A operand): row = 32jb + (lane&31), logical chunk = 2ks + hi
  const int kkey = (KCH == 16) ? (lq & 15) : ((lq >> 1) & 7);
  const unsigned k_rd = (unsigned)lq * KROW;
  // V^T frag (A operand of O^T = V^T · P^T): lane (d = 32dt + (lane & 31), hi) needs V[16kk + 8hi + i][d], i = 0..7, from
  // the row-major tile: two ds_read_b64_tr_b16 (i = 0..3, 4..7). In its 16-lane group g = lane >> 4 lane R = lane & 15
  // SUPPLIES the address of row 16kk + 8(g >> 1) + 4half + (R >> 2), d offset 32dt + 16(g & 1) + 4(R & 3) (4 elements)
  const int vg = lane >> 4, vR = lane & 15;
  const int v_r0 = 8 * (vg >> 1) + (vR >> 2);                                   // row of the lane within a 16-key step
  const int v_key = (KCH == 16) ? ((v_r0 & 3) << 2) : (((v_r0 >> 1) & 1) << 2); // (16kk + 4half) keeps these row bits
  const unsigned v_rd = (unsigned)v_r0 * KROW + (unsigned)(vR & 1) * 8u;
  unsigned v_col[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) v_col[dt] = (unsigned)(((4 * dt + 2 * (vg & 1) + ((vR & 3) >> 1)) ^ v_key) << 4);

  // all-ones A operand of the row-sum MFMA (OPT 4): 8 x 1.0 in the operand type
  vec8 ones8;
  {
    const unsigned one2 = pack2<TT>(1.0f, 1.0f);
    const u32x4_t o4 = {one2, one2, one2, one2};
    __builtin_memcpy(&ones8, &o4, 16);
  }

  // ---- S^T = K · Q^T of the K tile in buffer `buf` (a literal at every call site: offsets fold into the ds_read immediates).
  // The two key blocks accumulate alternately: no MFMA issues directly behind the one it depends on
  auto compute_s = [&](const int buf, f32x16_t (&s)[2]) {
    const unsigned char* sK = smem + buf * STAGE;
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[jb][r] = 0.f;
    if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int jb = 0; jb < 2; ++jb) {
        const vec8 kf = *(const vec8*)(sK + jb * 32 * KROW + k_rd + (((2 * ks + hi) ^ kkey) << 4));
        s[jb] = TT::mfma32(kf, qf[ks], s[jb]);
      }
    if (PRIO) __builtin_amdgcn_s_setprio(0);
  };

  // ---- online softmax of tile kvt (scores s) + O^T += V^T · P^T from the V tile in buffer `buf` ------------------------
  // (ILV: also S(t+1) of the K tile in buffer buf ^ 1 into sn, issued between the softmax's own instructions)
  auto softmax_pv = [&](const int buf, const int kvt, f32x16_t (&s)[2], f32x16_t (&sn)[2]) {
    const unsigned char* sV = smem + buf * STAGE + K_BYTES;
    const int kv0 = kvt * 64;
    // raw scores stay unscaled: p = exp2(s*c - m*c) is ONE fma + v_exp per element; masking only on edge tiles
    const float c = p.scale_log2;
    if (p.causal || kv0 + 64 > p.Skv) {  // wave-uniform: interior tiles skip the mask entirely
      const int kmax = p.causal ? (qrow + coff) : 0x7fffffff;
#pragma unroll
      for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kv = kv0 + 32 * jb + (r & 3) + 4 * ((r >> 2) & 1) + 8 * hi + 16 * (r >> 3);  // key of S^T row (see k_row)
          if (kv >= p.Skv || kv > kmax) s[jb][r] = -INFINITY;
        }
    }
    // the softmax is VALU-bound at head_dim 64 (32 exp + ~90 other VALU ops per lane per tile against 16 MFMAs): row max as
    // two v_max3 chains (16 ops, no canonicalising copies), row sum as one v_dot2 per packed pair (16 ops instead of 32 adds)
    float mloc = -INFINITY, mloc2 = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      mloc = max3f(mloc, s[0][r], s[1][r]);
      mloc2 = max3f(mloc2, s[0][r + 1], s[1][r + 1]);
    }
    if (SWAP) {
      // v_permlane32_swap vdst, src: lanes 32-63 of vdst <-> lanes 0-31 of src. With both = x: a = {x.lo, x.lo}, b = {x.hi, x.hi}.
      // (inline asm: hipcc 7.2 drops the builtin's second result; the s_nops cover the VALU-write → permlane-read wait states the
      // compiler cannot see into)
      float xa = fmaxf(mloc, mloc2), xb = xa;
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(xa), "+v"(xb));
      mloc = fmaxf(xa, xb);
    } else {
      mloc = max3f(mloc, mloc2, __shfl_xor(fmaxf(mloc, mloc2), 32, 64));
    }
    float alpha, mc;
    if (DEFER) {
      // deferred reference max (cdna_hip_programming.md T13): the exponent reference of a row moves only when the tile's maximum exceeds
      // it by more than 8 (log2 units), so P <= 2^8 instead of <= 1 (same relative rounding in fp16 / bf16; sums and O are fp32) and
      // the O rescale below leaves the per-tile path: with the exact running max some row of the wave's 32 raises its maximum in
      // most tiles (~135 updates per 64 tiles on the UNet's 4096-key rows), 16 v_pk_mul + the exp each time
      alpha = 1.0f;
      const bool need = (mloc - m_run) * c > 8.0f;   // -inf reference (first tile): +inf > 8; all-masked row: NaN > 8 is false
      if (__builtin_amdgcn_ballot_w64(need) != 0) {
        const float m_new = fmaxf(m_run, mloc);
        const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
        alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((m_run - m_safe) * c);
        m_run = m_new;
        mc_run = -m_safe * c;
      }
      mc = mc_run;
    } else {
      const float m_new = fmaxf(m_run, mloc);
      const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
      alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((m_run - m_safe) * c);
      m_run = m_new;
      mc = -m_safe * c;
    }
    float psum = 0.f;
    vec8 pb[4];
    // one softmax chunk = 4 scores → words 2*half, 2*half + 1 of pb[kk] (kk = 2jb + s2): 2 v_pk_fma, 4 v_exp, 2 cvt_pk, 2 dot2
    unsigned pw[4][4];
    auto chunk = [&](const int kk, const int half) {
      const int jb = kk >> 1, s2 = kk & 1;
#pragma unroll
      for (int j = 2 * half; j < 2 * half + 2; ++j) {
        typedef float f2_t __attribute__((ext_vector_type(2)));
        const f2_t sv = {s[jb][8 * s2 + 2 * j], s[jb][8 * s2 + 2 * j + 1]};
        const f2_t e = __builtin_elementwise_fma(sv, (f2_t){c, c}, (f2_t){mc, mc});
        const float p0 = __builtin_amdgcn_exp2f(e[0]);
        const float p1 = __builtin_amdgcn_exp2f(e[1]);
        pw[kk][j] = pack2<TT>(p0, p1);
        psum = TT::pair_sum(pw[kk][j], psum);
      }
    };
    if (!ILV)
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        unsigned w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // the two scores of a packed pair sit in an aligned register pair: one v_pk_fma_f32 scales and shifts both (same
          // IEEE fma per element, half the VALU issue slots)
          typedef float f2_t __attribute__((ext_vector_type(2)));
          const f2_t sv = {s[jb][8 * s2 + 2 * j], s[jb][8 * s2 + 2 * j + 1]};
          const f2_t e = __builtin_elementwise_fma(sv, (f2_t){c, c}, (f2_t){mc, mc});
          const float p0 = __builtin_amdgcn_exp2f(e[0]);
          const float p1 = __builtin_amdgcn_exp2f(e[1]);
          w[j] = pack2<TT>(p0, p1);
          if (!ONES) psum = TT::pair_sum(w[j], psum);          // sums the ROUNDED probabilities, i.e. exactly what P·V multiplies
        }
        __builtin_memcpy(&pb[2 * jb + s2], w, 16);
      }
    if (!ONES && !ILV) l_run = l_run * alpha + psum;
    if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {  // wave-uniform: the running max rarely moves after the first tiles
#pragma unroll
      for (int i = 0; i < NDT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
    }

    if (ILV) {
      // One basic block (it follows the rare O-rescale branch), issue order fixed by hand with sched_barrier fences:
      //   c0 S0 c1 S1 c2 S2 P0 c3 S3 P1 c4 S4 P2 c5 S5 P3 c6 S6 P4 c7 S7 P5 P6 P7
      // c = softmax chunk (4 scores: ~6 VALU + 4 v_exp ≈ 90 issue cycles), S = one MFMA of S(t+1) = K(t+1)·Qᵀ (unconditional: past
      // the last tile the buffer holds stale keys and the scores are never used), P = one MFMA of O += Vᵀ·Pᵀ, P(2kk + dt) as soon
      // as chunks 2kk, 2kk + 1 have produced pb[kk]. Every MFMA (32 matrix-pipe cycles) sits behind ≥ 1 chunk of independent VALU work, so
      // an in-order wave keeps its matrix pipe and its VALU busy at the same time; operand fragments are read from LDS one
      // step ahead. Same per-accumulator MFMA order as the round-3 kernel → bit-identical output.
      const unsigned char* sKn = smem + (buf ^ 1) * STAGE;
      auto read_k = [&](const int i) -> vec8 {          // i = 2ks + jb
        return *(const vec8*)(sKn + (i & 1) * 32 * KROW + k_rd + (((2 * (i >> 1) + hi) ^ kkey) << 4));
      };
      auto read_v = [&](const int pi) -> vec8 {         // pi = 2kk + dt (NDT == 2)
        const unsigned char* vp = sV + v_rd + v_col[pi & 1] + (pi >> 1) * 16 * KROW;
        const tr4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr4_t*)SX_LDS_PTR(vp));
        const tr4_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr4_t*)SX_LDS_PTR(vp + 4 * KROW));
        vec8 vf;
        __builtin_memcpy(&vf, &lo, 8);
        __builtin_memcpy((char*)&vf + 8, &hi4, 8);
        return vf;
      };
#pragma unroll
      for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int r = 0; r < 16; ++r) sn[jb][r] = 0.f;
      vec8 kfn = read_k(0), vfn;
      if (PRIO) __builtin_amdgcn_s_setprio(1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) {
        chunk(cc >> 1, cc & 1);
        if (cc & 1) __builtin_memcpy(&pb[cc >> 1], pw[cc >> 1], 16);
        __builtin_amdgcn_sched_barrier(0);
        sn[cc & 1] = TT::mfma32(kfn, qf[cc >> 1], sn[cc & 1]);
        if (cc + 1 < 8) kfn = read_k(cc + 1);
        if (cc >= 2) o[(cc - 2) & 1] = TT::mfma32(vfn, pb[(cc - 2) >> 1], o[(cc - 2) & 1]);
        if (cc >= 1) vfn = read_v(cc - 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      o[0] = TT::mfma32(vfn, pb[3], o[0]);
      vfn = read_v(7);
      o[1] = TT::mfma32(vfn, pb[3], o[1]);
      if (PRIO) __builtin_amdgcn_s_setprio(0);
      l_run = l_run * alpha + psum;
      return;
    }
    // ---- O^T += V^T · P^T (kk = 2jb + s2 : 16 keys; kk outer, d-tile inner: accumulators alternate) --------------------
    f32x16_t ls;
    if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (ONES) {
        if (kk == 0) {
          f32x16_t z;
#pragma unroll
          for (int r = 0; r < 16; ++r) z[r] = 0.f;
          ls = TT::mfma32(ones8, pb[0], z);
        } else {
          ls = TT::mfma32(ones8, pb[kk], ls);
        }
      }
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        const unsigned char* vp = sV + v_rd + v_col[dt] + kk * 16 * KROW;
        const tr4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr4_t*)SX_LDS_PTR(vp));
        const tr4_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr4_t*)SX_LDS_PTR(vp + 4 * KROW));
        vec8 vf;
        __builtin_memcpy(&vf, &lo, 8);
        __builtin_memcpy((char*)&vf + 8, &hi4, 8);
        o[dt] = TT::mfma32(vf, pb[kk], o[dt]);
      }
    }
    if (PRIO) __builtin_amdgcn_s_setprio(0);
    if (ONES) l_run = l_run * alpha + ls[0];       // every row of the ones-product is the full 64-key row sum (both lane halves)
  };

  f32x16_t sA[2], sB[2];
  if (!PIPE) {
    if (nt > 0) { stage_k(0, 0); stage_v(0, 0); }
    __syncthreads();
    // one KV tile; `cur` is a literal at both call sites
    auto tile = [&](const int cur, const int kvt) {
      if (kvt + 1 < nt) { stage_k(cur ^ 1, kvt + 1); stage_v(cur ^ 1, kvt + 1); }
      compute_s(cur, sA);
      softmax_pv(cur, kvt, sA, sB);
      __syncthreads();
    };
    for (int kvt = 0; kvt < nt; kvt += 2) {
      tile(0, kvt);
      if (kvt + 1 < nt) tile(1, kvt + 1);
    }
  } else {
    // K(t) lives in buffer t & 1 like V(t), but is staged one tile earlier: at tile t the DMA of K(t+2) and V(t+1) is issued, S(t+1)
    // is computed from K(t+1) (staged during tile t-1, complete since that tile's barrier) and the softmax + P·V of tile t run on
    // the scores computed one tile ago. A buffer's K half is overwritten one barrier after its last read (S(t) ran during tile
    // t-1), its V half likewise (P·V(t-1) ran during tile t-1).
    if (nt > 0) { stage_k(0, 0); stage_v(0, 0); }
    if (nt > 1) stage_k(1, 1);
    __syncthreads();
    if (nt > 0) compute_s(0, sA);
    for (int kvt = 0; kvt < nt; kvt += 2) {
      if (kvt + 2 < nt) stage_k(0, kvt + 2);
      if (kvt + 1 < nt) { stage_v(1, kvt + 1); if (!ILV) compute_s(1, sB); }
      softmax_pv(0, kvt, sA, sB);
      __syncthreads();
      if (kvt + 1 >= nt) break;
      if (kvt + 3 < nt) stage_k(1, kvt + 3);
      if (kvt + 2 < nt) { stage_v(0, kvt + 2); if (!ILV) compute_s(0, sA); }
      softmax_pv(1, kvt + 1, sB, sA);
      __syncthreads();
    }
  }

  // ---- epilogue: O[q][d] = o / l ------------------------------------------------------------------------
  const float l_tot = ONES ? l_run : l_run + __shfl_xor(l_run, 32, 64);
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
  if (qrow < p.Sq) {
    unsigned short* Or = p.O + b * p.obs + (long long)qrow * p.ors + (long long)h * p.D;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = 32 * dt + 8 * g + 4 * hi;
        if (d < p.D) {
          u32x2_t w;
          w[0] = pack2<TT>(o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv);
          w[1] = pack2<TT>(o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
          *(u32x2_t*)(Or + d) = w;
        }
      }
  }
#endif
}

// ---- head_dim-64 kernel, ONE wave per SIMD: 4 waves x 64 query rows (two 32-row q-blocks a, b per wave) ---------------------
// Round-6 experiment for the UNet's self-attention. Same tile math, layouts and softmax formula as attn_kernel
// <.., 64, ..> above; what changes is who overlaps with whom. attn_kernel runs 3-4 workgroups per CU and leaves the overlap of one
// wave's MFMAs with another wave's softmax to the SIMD arbiter (PMC: matrix 34-42 % + VALU 59-67 % ~ 100 %: they alternate). Here a
// workgroup owns the CU (launch bounds (256, 1): the whole 512-register file per wave) and every wave's own instruction stream
// carries an MFMA every ~5-6 VALU issues:
//   * S(t+1) = K(t+1) Q^T is computed one tile ahead into a second score register set (sn), so the MFMAs issued while the softmax of
//     tile t runs are independent of it
//   * the two q-blocks share every K fragment (ds_read_b128) and every V^T fragment (2 ds_read_b64_tr_b16): half the LDS reads per
//     MFMA of attn_kernel; one barrier per 64 keys for 256 query rows instead of 128
//   * O^T (and, OPT 1, the row sums as a third "d tile" against an all-ones A operand) accumulate in AGPRs through inline-asm MFMAs:
//     the arch VGPRs hold s, sn, Q fragments and P
//   * OPT 2: the last 16-key step of P·V of tile t is issued at the head of tile t+1 (its V tile stays valid: 4-slot ring), so no
//     tile ends in an MFMA-only tail
//   * K/V tiles by LDS-DMA into a 4-slot ring, K three tiles ahead and V two, retired by a counted s_waitcnt vmcnt(4) (this
//     tile's own four DMA instructions stay in flight across the barrier); tiles past the end are fetched as zeros through the
//     buffer range check, so the count has no edge cases
// Not causal; head_dim <= 64 (zero-padded like attn_kernel).
// The O^T accumulators of attn64_kernel live in a[0:95], OUTSIDE the compiler's view: tuple T = 3 x + i (q-block x; i = 0, 1: d tiles, 2: row
// sums) is a[16 T : 16 T + 15], named literally in every asm statement that touches it. As C++ values bound to "+a" operands they
// cost 192 v_accvgpr moves per tile: the conditional rescale makes them phis, phis of 16-float vectors are VGPR-class, and every asm
// MFMA gets its accumulator copied in. Every statement clobbers all 96 registers, so the allocator never keeps anything there across
// one of them (and the tile loop has one every few instructions); the build is checked for compiler-generated uses of a0-a95
// (tests/test_cpu_suite.py::test_attn64_accumulators_are_private).
#define SX_ACC_CLOBBERS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95"
template <typename TT, bool WAIT_LDS = false>
__device__ __forceinline__ void acc_mfma(const int T, const typename TT::vec8& a, const typename TT::vec8& b) {
#if defined(__HIP_DEVICE_COMPILE__)
  // s_nop 1: VALU write (the packed P) → MFMA read wait states the compiler cannot insert for an asm consumer; WAIT_LDS: the fragment
  // came from an inline-asm ds_read the compiler does not count
#define SX_MF_F16 "v_mfma_f32_32x32x16_f16"
#define SX_MF_BF16 "v_mfma_f32_32x32x16_bf16"
  if constexpr (std::is_same<TT, F16>::value) {
#define SX_MF SX_MF_F16
    switch (T) {
    case 0:
      if (WAIT_LDS) asm volatile("s_waitcnt lgkmcnt(0)\n\t" SX_MF " a[0:15], %0, %1, a[0:15]" ::"v"(a), "v"(b) : SX_ACC_CLOBBERS);
      else asm volatile("s_nop 1\n\t" SX_MF " a[0:15], %0, %1, a[0:15]" ::"v"(a), "v"(b) : SX_ACC_CLOBBERS);
      break;
    case 1:
      if (WAIT_LDS) asm volatile("s_waitcnt lgkmcnt(0)\n\t" SX_MF " a[16:31], %0, %1, a[16:31]" ::"v"(a), "v"(b) : SX_ACC_CLOBBERS);
      else asm volatile("s_nop 1\n\t" SX_MF " a[16:31], %0, %1, a[16:31]" ::"v"(a), "v"(b) : SX_ACC_CLOBBERS);
      break;
    case 2:
      if (WAIT_LDS) asm volatile("s_waitcnt lgkmcnt(0)\n\t" SX_MF " a[32:47], %0, %1, a[32:47]" ::"v"(a), "v"(b) : SX_ACC_CLOBBERS);
      else asm volatile("s_nop 1\n\t" SX_MF " a[32:47], %0, %1, a[32:47]" ::"v"(a), "v"(b) : SX_ACC_CLOBBERS);
      break;
    case 3:
      if (WAIT_LDS) asm volatile("s_waitcnt lgkmcnt(0)\n\t" SX_MF " a[48:63], %0, %1, a[48:63]" ::"v"(a), "v"(b) : SX_ACC_CLOBBERS);
      else asm volatile("s_nop 1\n\t" SX_MF " a[48:63], %0, %1, a[48:63]" ::"v"(a), "v"(b) : SX_ACC_CLOBBERS);
      break;
    case 4:
      if (WAIT_LDS) asm volatile("s_waitcnt lgkmcnt(0)\n\t" SX_MF " a[64:79], %0, %1, a[64:79]" ::"v"(a), "v"(b) : SX_ACC_CLOBBERS);
      else asm volatile("s_nop 1\n\t" SX_MF " a[64:79], %0, %1, a[64:79]" ::"v"(a), "v"(b) : SX_ACC_CLOBBERS);
      break;
    case 5:
      if (WAIT_LDS) asm volatile("s_waitcnt lgkmcnt(0)\n\t" SX_MF " a[80:95], %0, %1, a[80:95]" ::"v"(a), "v"(b) : SX_ACC_CLOBBERS);
      else asm volatile("s_nop 1\n\t" SX_MF " a[80:95], %0, %1, a[80:95]" ::"v"(a), "v"(b) : SX_ACC_CLOBBERS);
      break;
    }
#undef SX_MF
  } else {
#define SX_MF SX_MF_BF16
    switch (T) {
    case 0:
      if (WAIT_LDS) asm volatile("s_waitcnt lgkmcnt(0)\n\t" SX_MF " a[0:15], %0, %1, a[0:15]" ::"v"(a), "v"(b) : SX_ACC_CLOBBERS);
      else asm volatile("s_nop 1\n\t" SX_MF " a[0:15], %0, %1, a[0:15]" ::"v"(a), "v"(b) : SX_ACC_CLOBBERS);
      break;
    case 1:
      if (WAIT_LDS) asm volatile("s_waitcnt lgkmcnt(0)\n\t" SX_MF " a[16:31], %0, %1, a[16:31]" ::"v"(a), "v"(b) : SX_ACC_CLOBBERS);
      else asm volatile("s_nop 1\n\t" SX_MF " a[16:31], %0, %1, a[16:31]" ::"v"(a), "v"(b) : SX_ACC_CLOBBERS);
      break;
    case 2:
      if (WAIT_LDS) asm volatile("s_waitcnt lgkmcnt(0)\n\t" SX_MF " a[32:47], %0, %1, a[32:47]" ::"v"(a), "v"(b) : SX_ACC_CLOBBERS);
      else asm volatile("s_nop 1\n\t" SX_MF " a[32:47], %0, %1, a[32:47]" ::"v"(a), "v"(b) : SX_ACC_CLOBBERS);
      break;
    case 3:
      if (WAIT_LDS) asm volatile("s_waitcnt lgkmcnt(0)\n\t" SX_MF " a[48:63], %0, %1, a[48:63]" ::"v"(a), "v"(b) : SX_ACC_CLOBBERS);
      else asm volatile("s_nop 1\n\t" SX_MF " a[48:63], %0, %1, a[48:63]" ::"v"(a), "v"(b) : SX_ACC_CLOBBERS);
      break;
    case 4:
      if (WAIT_LDS) asm volatile("s_waitcnt lgkmcnt(0)\n\t" SX_MF " a[64:79], %0, %1, a[64:79]" ::"v"(a), "v"(b) : SX_ACC_CLOBBERS);
      else asm volatile("s_nop 1\n\t" SX_MF " a[64:79], %0, %1, a[64:79]" ::"v"(a), "v"(b) : SX_ACC_CLOBBERS);
      break;
    case 5:
      if (WAIT_LDS) asm volatile("s_waitcnt lgkmcnt(0)\n\t" SX_MF " a[80:95], %0, %1, a[80:95]" ::"v"(a), "v"(b) : SX_ACC_CLOBBERS);
      else asm volatile("s_nop 1\n\t" SX_MF " a[80:95], %0, %1, a[80:95]" ::"v"(a), "v"(b) : SX_ACC_CLOBBERS);
      break;
    }
#undef SX_MF
  }
#endif
}
__device__ __forceinline__ void acc_scale(const int T, const float f) {   // tuple T *= f (the rare running-max update)
#if defined(__HIP_DEVICE_COMPILE__)
  float t;
  switch (T) {
    case 0: asm volatile("s_nop 7\n\tv_accvgpr_read_b32 %0, a0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a0, %0\n\tv_accvgpr_read_b32 %0, a1\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a1, %0\n\tv_accvgpr_read_b32 %0, a2\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a2, %0\n\tv_accvgpr_read_b32 %0, a3\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a3, %0\n\tv_accvgpr_read_b32 %0, a4\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a4, %0\n\tv_accvgpr_read_b32 %0, a5\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a5, %0\n\tv_accvgpr_read_b32 %0, a6\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a6, %0\n\tv_accvgpr_read_b32 %0, a7\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a7, %0\n\tv_accvgpr_read_b32 %0, a8\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a8, %0\n\tv_accvgpr_read_b32 %0, a9\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a9, %0\n\tv_accvgpr_read_b32 %0, a10\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a10, %0\n\tv_accvgpr_read_b32 %0, a11\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a11, %0\n\tv_accvgpr_read_b32 %0, a12\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a12, %0\n\tv_accvgpr_read_b32 %0, a13\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a13, %0\n\tv_accvgpr_read_b32 %0, a14\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a14, %0\n\tv_accvgpr_read_b32 %0, a15\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a15, %0\n\ts_nop 3" : "=&v"(t) : "v"(f) : SX_ACC_CLOBBERS); break;
    case 1: asm volatile("s_nop 7\n\tv_accvgpr_read_b32 %0, a16\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a16, %0\n\tv_accvgpr_read_b32 %0, a17\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a17, %0\n\tv_accvgpr_read_b32 %0, a18\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a18, %0\n\tv_accvgpr_read_b32 %0, a19\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a19, %0\n\tv_accvgpr_read_b32 %0, a20\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a20, %0\n\tv_accvgpr_read_b32 %0, a21\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a21, %0\n\tv_accvgpr_read_b32 %0, a22\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a22, %0\n\tv_accvgpr_read_b32 %0, a23\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a23, %0\n\tv_accvgpr_read_b32 %0, a24\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a24, %0\n\tv_accvgpr_read_b32 %0, a25\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a25, %0\n\tv_accvgpr_read_b32 %0, a26\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a26, %0\n\tv_accvgpr_read_b32 %0, a27\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a27, %0\n\tv_accvgpr_read_b32 %0, a28\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a28, %0\n\tv_accvgpr_read_b32 %0, a29\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a29, %0\n\tv_accvgpr_read_b32 %0, a30\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a30, %0\n\tv_accvgpr_read_b32 %0, a31\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a31, %0\n\ts_nop 3" : "=&v"(t) : "v"(f) : SX_ACC_CLOBBERS); break;
    case 2: asm volatile("s_nop 7\n\tv_accvgpr_read_b32 %0, a32\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a32, %0\n\tv_accvgpr_read_b32 %0, a33\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a33, %0\n\tv_accvgpr_read_b32 %0, a34\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a34, %0\n\tv_accvgpr_read_b32 %0, a35\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a35, %0\n\tv_accvgpr_read_b32 %0, a36\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a36, %0\n\tv_accvgpr_read_b32 %0, a37\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a37, %0\n\tv_accvgpr_read_b32 %0, a38\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a38, %0\n\tv_accvgpr_read_b32 %0, a39\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a39, %0\n\tv_accvgpr_read_b32 %0, a40\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a40, %0\n\tv_accvgpr_read_b32 %0, a41\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a41, %0\n\tv_accvgpr_read_b32 %0, a42\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a42, %0\n\tv_accvgpr_read_b32 %0, a43\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a43, %0\n\tv_accvgpr_read_b32 %0, a44\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a44, %0\n\tv_accvgpr_read_b32 %0, a45\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a45, %0\n\tv_accvgpr_read_b32 %0, a46\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a46, %0\n\tv_accvgpr_read_b32 %0, a47\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a47, %0\n\ts_nop 3" : "=&v"(t) : "v"(f) : SX_ACC_CLOBBERS); break;
    case 3: asm volatile("s_nop 7\n\tv_accvgpr_read_b32 %0, a48\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a48, %0\n\tv_accvgpr_read_b32 %0, a49\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a49, %0\n\tv_accvgpr_read_b32 %0, a50\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a50, %0\n\tv_accvgpr_read_b32 %0, a51\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a51, %0\n\tv_accvgpr_read_b32 %0, a52\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a52, %0\n\tv_accvgpr_read_b32 %0, a53\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a53, %0\n\tv_accvgpr_read_b32 %0, a54\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a54, %0\n\tv_accvgpr_read_b32 %0, a55\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a55, %0\n\tv_accvgpr_read_b32 %0, a56\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a56, %0\n\tv_accvgpr_read_b32 %0, a57\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a57, %0\n\tv_accvgpr_read_b32 %0, a58\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a58, %0\n\tv_accvgpr_read_b32 %0, a59\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a59, %0\n\tv_accvgpr_read_b32 %0, a60\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a60, %0\n\tv_accvgpr_read_b32 %0, a61\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a61, %0\n\tv_accvgpr_read_b32 %0, a62\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a62, %0\n\tv_accvgpr_read_b32 %0, a63\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a63, %0\n\ts_nop 3" : "=&v"(t) : "v"(f) : SX_ACC_CLOBBERS); break;
    case 4: asm volatile("s_nop 7\n\tv_accvgpr_read_b32 %0, a64\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a64, %0\n\tv_accvgpr_read_b32 %0, a65\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a65, %0\n\tv_accvgpr_read_b32 %0, a66\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a66, %0\n\tv_accvgpr_read_b32 %0, a67\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a67, %0\n\tv_accvgpr_read_b32 %0, a68\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a68, %0\n\tv_accvgpr_read_b32 %0, a69\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a69, %0\n\tv_accvgpr_read_b32 %0, a70\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a70, %0\n\tv_accvgpr_read_b32 %0, a71\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a71, %0\n\tv_accvgpr_read_b32 %0, a72\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a72, %0\n\tv_accvgpr_read_b32 %0, a73\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a73, %0\n\tv_accvgpr_read_b32 %0, a74\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a74, %0\n\tv_accvgpr_read_b32 %0, a75\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a75, %0\n\tv_accvgpr_read_b32 %0, a76\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a76, %0\n\tv_accvgpr_read_b32 %0, a77\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a77, %0\n\tv_accvgpr_read_b32 %0, a78\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a78, %0\n\tv_accvgpr_read_b32 %0, a79\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a79, %0\n\ts_nop 3" : "=&v"(t) : "v"(f) : SX_ACC_CLOBBERS); break;
    case 5: asm volatile("s_nop 7\n\tv_accvgpr_read_b32 %0, a80\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a80, %0\n\tv_accvgpr_read_b32 %0, a81\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a81, %0\n\tv_accvgpr_read_b32 %0, a82\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a82, %0\n\tv_accvgpr_read_b32 %0, a83\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a83, %0\n\tv_accvgpr_read_b32 %0, a84\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a84, %0\n\tv_accvgpr_read_b32 %0, a85\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a85, %0\n\tv_accvgpr_read_b32 %0, a86\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a86, %0\n\tv_accvgpr_read_b32 %0, a87\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a87, %0\n\tv_accvgpr_read_b32 %0, a88\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a88, %0\n\tv_accvgpr_read_b32 %0, a89\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a89, %0\n\tv_accvgpr_read_b32 %0, a90\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a90, %0\n\tv_accvgpr_read_b32 %0, a91\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a91, %0\n\tv_accvgpr_read_b32 %0, a92\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a92, %0\n\tv_accvgpr_read_b32 %0, a93\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a93, %0\n\tv_accvgpr_read_b32 %0, a94\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a94, %0\n\tv_accvgpr_read_b32 %0, a95\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a95, %0\n\ts_nop 3" : "=&v"(t) : "v"(f) : SX_ACC_CLOBBERS); break;
  }
  (void)t;
#endif
}
__device__ __forceinline__ void acc_zero() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("v_accvgpr_write_b32 a0, 0\n\tv_accvgpr_write_b32 a1, 0\n\tv_accvgpr_write_b32 a2, 0\n\tv_accvgpr_write_b32 a3, 0\n\tv_accvgpr_write_b32 a4, 0\n\tv_accvgpr_write_b32 a5, 0\n\tv_accvgpr_write_b32 a6, 0\n\tv_accvgpr_write_b32 a7, 0\n\tv_accvgpr_write_b32 a8, 0\n\tv_accvgpr_write_b32 a9, 0\n\tv_accvgpr_write_b32 a10, 0\n\tv_accvgpr_write_b32 a11, 0\n\tv_accvgpr_write_b32 a12, 0\n\tv_accvgpr_write_b32 a13, 0\n\tv_accvgpr_write_b32 a14, 0\n\tv_accvgpr_write_b32 a15, 0\n\tv_accvgpr_write_b32 a16, 0\n\tv_accvgpr_write_b32 a17, 0\n\tv_accvgpr_write_b32 a18, 0\n\tv_accvgpr_write_b32 a19, 0\n\tv_accvgpr_write_b32 a20, 0\n\tv_accvgpr_write_b32 a21, 0\n\tv_accvgpr_write_b32 a22, 0\n\tv_accvgpr_write_b32 a23, 0\n\tv_accvgpr_write_b32 a24, 0\n\tv_accvgpr_write_b32 a25, 0\n\tv_accvgpr_write_b32 a26, 0\n\tv_accvgpr_write_b32 a27, 0\n\tv_accvgpr_write_b32 a28, 0\n\tv_accvgpr_write_b32 a29, 0\n\tv_accvgpr_write_b32 a30, 0\n\tv_accvgpr_write_b32 a31, 0\n\tv_accvgpr_write_b32 a32, 0\n\tv_accvgpr_write_b32 a33, 0\n\tv_accvgpr_write_b32 a34, 0\n\tv_accvgpr_write_b32 a35, 0\n\tv_accvgpr_write_b32 a36, 0\n\tv_accvgpr_write_b32 a37, 0\n\tv_accvgpr_write_b32 a38, 0\n\tv_accvgpr_write_b32 a39, 0\n\tv_accvgpr_write_b32 a40, 0\n\tv_accvgpr_write_b32 a41, 0\n\tv_accvgpr_write_b32 a42, 0\n\tv_accvgpr_write_b32 a43, 0\n\tv_accvgpr_write_b32 a44, 0\n\tv_accvgpr_write_b32 a45, 0\n\tv_accvgpr_write_b32 a46, 0\n\tv_accvgpr_write_b32 a47, 0\n\tv_accvgpr_write_b32 a48, 0\n\tv_accvgpr_write_b32 a49, 0\n\tv_accvgpr_write_b32 a50, 0\n\tv_accvgpr_write_b32 a51, 0\n\tv_accvgpr_write_b32 a52, 0\n\tv_accvgpr_write_b32 a53, 0\n\tv_accvgpr_write_b32 a54, 0\n\tv_accvgpr_write_b32 a55, 0\n\tv_accvgpr_write_b32 a56, 0\n\tv_accvgpr_write_b32 a57, 0\n\tv_accvgpr_write_b32 a58, 0\n\tv_accvgpr_write_b32 a59, 0\n\tv_accvgpr_write_b32 a60, 0\n\tv_accvgpr_write_b32 a61, 0\n\tv_accvgpr_write_b32 a62, 0\n\tv_accvgpr_write_b32 a63, 0\n\tv_accvgpr_write_b32 a64, 0\n\tv_accvgpr_write_b32 a65, 0\n\tv_accvgpr_write_b32 a66, 0\n\tv_accvgpr_write_b32 a67, 0\n\tv_accvgpr_write_b32 a68, 0\n\tv_accvgpr_write_b32 a69, 0\n\tv_accvgpr_write_b32 a70, 0\n\tv_accvgpr_write_b32 a71, 0\n\tv_accvgpr_write_b32 a72, 0\n\tv_accvgpr_write_b32 a73, 0\n\tv_accvgpr_write_b32 a74, 0\n\tv_accvgpr_write_b32 a75, 0\n\tv_accvgpr_write_b32 a76, 0\n\tv_accvgpr_write_b32 a77, 0\n\tv_accvgpr_write_b32 a78, 0\n\tv_accvgpr_write_b32 a79, 0\n\tv_accvgpr_write_b32 a80, 0\n\tv_accvgpr_write_b32 a81, 0\n\tv_accvgpr_write_b32 a82, 0\n\tv_accvgpr_write_b32 a83, 0\n\tv_accvgpr_write_b32 a84, 0\n\tv_accvgpr_write_b32 a85, 0\n\tv_accvgpr_write_b32 a86, 0\n\tv_accvgpr_write_b32 a87, 0\n\tv_accvgpr_write_b32 a88, 0\n\tv_accvgpr_write_b32 a89, 0\n\tv_accvgpr_write_b32 a90, 0\n\tv_accvgpr_write_b32 a91, 0\n\tv_accvgpr_write_b32 a92, 0\n\tv_accvgpr_write_b32 a93, 0\n\tv_accvgpr_write_b32 a94, 0\n\tv_accvgpr_write_b32 a95, 0\n\ts_nop 3" ::: SX_ACC_CLOBBERS);
#endif
}
__device__ __forceinline__ void acc_read(const int T, float (&v)[16]) {
#if defined(__HIP_DEVICE_COMPILE__)
  switch (T) {
    case 0:
      asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(v[0]));
      asm volatile("v_accvgpr_read_b32 %0, a1" : "=v"(v[1]));
      asm volatile("v_accvgpr_read_b32 %0, a2" : "=v"(v[2]));
      asm volatile("v_accvgpr_read_b32 %0, a3" : "=v"(v[3]));
      asm volatile("v_accvgpr_read_b32 %0, a4" : "=v"(v[4]));
      asm volatile("v_accvgpr_read_b32 %0, a5" : "=v"(v[5]));
      asm volatile("v_accvgpr_read_b32 %0, a6" : "=v"(v[6]));
      asm volatile("v_accvgpr_read_b32 %0, a7" : "=v"(v[7]));
      asm volatile("v_accvgpr_read_b32 %0, a8" : "=v"(v[8]));
      asm volatile("v_accvgpr_read_b32 %0, a9" : "=v"(v[9]));
      asm volatile("v_accvgpr_read_b32 %0, a10" : "=v"(v[10]));
      asm volatile("v_accvgpr_read_b32 %0, a11" : "=v"(v[11]));
      asm volatile("v_accvgpr_read_b32 %0, a12" : "=v"(v[12]));
      asm volatile("v_accvgpr_read_b32 %0, a13" : "=v"(v[13]));
      asm volatile("v_accvgpr_read_b32 %0, a14" : "=v"(v[14]));
      asm volatile("v_accvgpr_read_b32 %0, a15" : "=v"(v[15]));
      break;
    case 1:
      asm volatile("v_accvgpr_read_b32 %0, a16" : "=v"(v[0]));
      asm volatile("v_accvgpr_read_b32 %0, a17" : "=v"(v[1]));
      asm volatile("v_accvgpr_read_b32 %0, a18" : "=v"(v[2]));
      asm volatile("v_accvgpr_read_b32 %0, a19" : "=v"(v[3]));
      asm volatile("v_accvgpr_read_b32 %0, a20" : "=v"(v[4]));
      asm volatile("v_accvgpr_read_b32 %0, a21" : "=v"(v[5]));
      asm volatile("v_accvgpr_read_b32 %0, a22" : "=v"(v[6]));
      asm volatile("v_accvgpr_read_b32 %0, a23" : "=v"(v[7]));
      asm volatile("v_accvgpr_read_b32 %0, a24" : "=v"(v[8]));
      asm volatile("v_accvgpr_read_b32 %0, a25" : "=v"(v[9]));
      asm volatile("v_accvgpr_read_b32 %0, a26" : "=v"(v[10]));
      asm volatile("v_accvgpr_read_b32 %0, a27" : "=v"(v[11]));
      asm volatile("v_accvgpr_read_b32 %0, a28" : "=v"(v[12]));
      asm volatile("v_accvgpr_read_b32 %0, a29" : "=v"(v[13]));
      asm volatile("v_accvgpr_read_b32 %0, a30" : "=v"(v[14]));
      asm volatile("v_accvgpr_read_b32 %0, a31" : "=v"(v[15]));
      break;
    case 2:
      asm volatile("v_accvgpr_read_b32 %0, a32" : "=v"(v[0]));
      asm volatile("v_accvgpr_read_b32 %0, a33" : "=v"(v[1]));
      asm volatile("v_accvgpr_read_b32 %0, a34" : "=v"(v[2]));
      asm volatile("v_accvgpr_read_b32 %0, a35" : "=v"(v[3]));
      asm volatile("v_accvgpr_read_b32 %0, a36" : "=v"(v[4]));
      asm volatile("v_accvgpr_read_b32 %0, a37" : "=v"(v[5]));
      asm volatile("v_accvgpr_read_b32 %0, a38" : "=v"(v[6]));
      asm volatile("v_accvgpr_read_b32 %0, a39" : "=v"(v[7]));
      asm volatile("v_accvgpr_read_b32 %0, a40" : "=v"(v[8]));
      asm volatile("v_accvgpr_read_b32 %0, a41" : "=v"(v[9]));
      asm volatile("v_accvgpr_read_b32 %0, a42" : "=v"(v[10]));
      asm volatile("v_accvgpr_read_b32 %0, a43" : "=v"(v[11]));
      asm volatile("v_accvgpr_read_b32 %0, a44" : "=v"(v[12]));
      asm volatile("v_accvgpr_read_b32 %0, a45" : "=v"(v[13]));
      asm volatile("v_accvgpr_read_b32 %0, a46" : "=v"(v[14]));
      asm volatile("v_accvgpr_read_b32 %0, a47" : "=v"(v[15]));
      break;
    case 3:
      asm volatile("v_accvgpr_read_b32 %0, a48" : "=v"(v[0]));
      asm volatile("v_accvgpr_read_b32 %0, a49" : "=v"(v[1]));
      asm volatile("v_accvgpr_read_b32 %0, a50" : "=v"(v[2]));
      asm volatile("v_accvgpr_read_b32 %0, a51" : "=v"(v[3]));
      asm volatile("v_accvgpr_read_b32 %0, a52" : "=v"(v[4]));
      asm volatile("v_accvgpr_read_b32 %0, a53" : "=v"(v[5]));
      asm volatile("v_accvgpr_read_b32 %0, a54" : "=v"(v[6]));
      asm volatile("v_accvgpr_read_b32 %0, a55" : "=v"(v[7]));
      asm volatile("v_accvgpr_read_b32 %0, a56" : "=v"(v[8]));
      asm volatile("v_accvgpr_read_b32 %0, a57" : "=v"(v[9]));
      asm volatile("v_accvgpr_read_b32 %0, a58" : "=v"(v[10]));
      asm volatile("v_accvgpr_read_b32 %0, a59" : "=v"(v[11]));
      asm volatile("v_accvgpr_read_b32 %0, a60" : "=v"(v[12]));
      asm volatile("v_accvgpr_read_b32 %0, a61" : "=v"(v[13]));
      asm volatile("v_accvgpr_read_b32 %0, a62" : "=v"(v[14]));
      asm volatile("v_accvgpr_read_b32 %0, a63" : "=v"(v[15]));
      break;
    case 4:
      asm volatile("v_accvgpr_read_b32 %0, a64" : "=v"(v[0]));
      asm volatile("v_accvgpr_read_b32 %0, a65" : "=v"(v[1]));
      asm volatile("v_accvgpr_read_b32 %0, a66" : "=v"(v[2]));
      asm volatile("v_accvgpr_read_b32 %0, a67" : "=v"(v[3]));
      asm volatile("v_accvgpr_read_b32 %0, a68" : "=v"(v[4]));
      asm volatile("v_accvgpr_read_b32 %0, a69" : "=v"(v[5]));
      asm volatile("v_accvgpr_read_b32 %0, a70" : "=v"(v[6]));
      asm volatile("v_accvgpr_read_b32 %0, a71" : "=v"(v[7]));
      asm volatile("v_accvgpr_read_b32 %0, a72" : "=v"(v[8]));
      asm volatile("v_accvgpr_read_b32 %0, a73" : "=v"(v[9]));
      asm volatile("v_accvgpr_read_b32 %0, a74" : "=v"(v[10]));
      asm volatile("v_accvgpr_read_b32 %0, a75" : "=v"(v[11]));
      asm volatile("v_accvgpr_read_b32 %0, a76" : "=v"(v[12]));
      asm volatile("v_accvgpr_read_b32 %0, a77" : "=v"(v[13]));
      asm volatile("v_accvgpr_read_b32 %0, a78" : "=v"(v[14]));
      asm volatile("v_accvgpr_read_b32 %0, a79" : "=v"(v[15]));
      break;
    case 5:
      asm volatile("v_accvgpr_read_b32 %0, a80" : "=v"(v[0]));
      asm volatile("v_accvgpr_read_b32 %0, a81" : "=v"(v[1]));
      asm volatile("v_accvgpr_read_b32 %0, a82" : "=v"(v[2]));
      asm volatile("v_accvgpr_read_b32 %0, a83" : "=v"(v[3]));
      asm volatile("v_accvgpr_read_b32 %0, a84" : "=v"(v[4]));
      asm volatile("v_accvgpr_read_b32 %0, a85" : "=v"(v[5]));
      asm volatile("v_accvgpr_read_b32 %0, a86" : "=v"(v[6]));
      asm volatile("v_accvgpr_read_b32 %0, a87" : "=v"(v[7]));
      asm volatile("v_accvgpr_read_b32 %0, a88" : "=v"(v[8]));
      asm volatile("v_accvgpr_read_b32 %0, a89" : "=v"(v[9]));
      asm volatile("v_accvgpr_read_b32 %0, a90" : "=v"(v[10]));
      asm volatile("v_accvgpr_read_b32 %0, a91" : "=v"(v[11]));
      asm volatile("v_accvgpr_read_b32 %0, a92" : "=v"(v[12]));
      asm volatile("v_accvgpr_read_b32 %0, a93" : "=v"(v[13]));
      asm volatile("v_accvgpr_read_b32 %0, a94" : "=v"(v[14]));
      asm volatile("v_accvgpr_read_b32 %0, a95" : "=v"(v[15]));
      break;
  }
#endif
}

template <typename TT, int OPT>
__global__ __launch_bounds__(256, 1) void attn64_kernel(const AttnP p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef typename TT::vec8 vec8;
  constexpr bool ONES = true, CARRY = (OPT & 2) != 0, SCALAR = (OPT & 4) != 0, PRIO = (OPT & 8) != 0;   // OPT bit 1: deferred reference max (THR = 8)
  // timing ablations (wrong results; tools/lab/attn_lab prints the times only): 16 no v_exp, 32 no P·V / row-sum MFMAs, 64 no S MFMAs,
  // 128 no K/V DMA in the tile loop, 256 no softmax VALU at all
  constexpr bool PROBE = (OPT & 512) != 0;   // s_memtime stamps per tile → O[0..15] of workgroup 0 (cycles: DMA issue | compute | vmcnt wait | barrier)
  constexpr bool AB_EXP = (OPT & 16) != 0, AB_PV = (OPT & 32) != 0, AB_S = (OPT & 64) != 0, AB_DMA = (OPT & 128) != 0, AB_SM = (OPT & 256) != 0;
  constexpr int KROW = 128, K_BYTES = 64 * KROW, STAGE = 2 * K_BYTES, NSLOT = 8;   // ring: K(t) and V(t) live in slot t & 7
  constexpr int NDT = ONES ? 3 : 2;          // O^T d tiles per q-block (+ the row-sum tile)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, lq = lane & 31;
  const int t = xcd_remap(blockIdx.x, gridDim.x);
  const int qt = t % p.q_tiles, bh = t / p.q_tiles;
  const int b = bh / p.H, h = bh % p.H;
  const int q0 = qt * 256 + wave * 64;
  const int nt = (p.Skv + 63) / 64;

  const unsigned short* Kb = p.K + b * p.kbs + h * p.khs;
  const unsigned short* Vb = p.V + b * p.vbs + h * p.vhs;
  const int k_bytes = (int)(((long long)(p.Skv - 1) * p.krs + p.D) * 2);
  const int v_bytes = (int)(((long long)(p.Skv - 1) * p.vrs + p.D) * 2);
  __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc((void*)Kb, 0, k_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc((void*)Vb, 0, v_bytes, 0x00020000);

  // per-lane DMA sources: 8 one-KiB slots per tile (8 rows of 128 B), two per wave; see attn_kernel for the row order / swizzles
  unsigned k_off[2], v_off[2];
  int k_row[2], v_row[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (wave * 2 + i) * 8 + (lane >> 3);
    const int pos = lane & 7;
    const int ck = pos ^ ((r >> 1) & 7), cv = pos ^ (((r >> 1) & 1) << 2);
    k_row[i] = (r & ~0xC) | ((r & 4) << 1) | ((r & 8) >> 1);
    k_off[i] = (ck * 8 < p.D) ? (unsigned)(ck * 16) : 0x80000000u;
    v_row[i] = r;
    v_off[i] = (cv * 8 < p.D) ? (unsigned)(cv * 16) : 0x80000000u;
  }
  auto stage_k = [&](const int slot, const int kvt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int kv = kvt * 64 + k_row[i];
      const unsigned voff = (kv < p.Skv) ? (unsigned)((long long)kv * p.krs * 2) + k_off[i] : 0x80000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rK, SX_LDS_PTR(smem + slot * STAGE + (wave * 2 + i) * 1024), 16, voff, 0, 0, 0);
    }
  };
  auto stage_v = [&](const int slot, const int kvt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int kv = kvt * 64 + v_row[i];
      const unsigned voff = (kv < p.Skv) ? (unsigned)((long long)kv * p.vrs * 2) + v_off[i] : 0x80000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rV, SX_LDS_PTR(smem + slot * STAGE + K_BYTES + (wave * 2 + i) * 1024), 16, voff, 0, 0,
                                               0);
    }
  };

  // prologue DMA in the steady-state order: K0 | K1 V0 | K2 V1 | .. | K6 V5 (+ zeros into V slot 7 for the carried P·V of "tile -1").
  // One workgroup per CU means nobody else's loads cover this workgroup's HBM/L2 round trip (measured here ~2.3 us under load, lab
  // ablation: DMA + LDS reads alone ran at 14 GB/s per CU with 32 KB in flight): 6 tiles (96 KB) stay in flight instead
  stage_k(0, 0);
#pragma unroll
  for (int i = 0; i < NSLOT - 2; ++i) { stage_k(i + 1, i + 1); stage_v(i, i); }
  if (CARRY) stage_v(NSLOT - 1, nt + NSLOT);   // past the end: zero fill
  constexpr int PRO_DMA = 2 + 4 * (NSLOT - 2) + (CARRY ? 2 : 0);

  // Q fragments (B operand of S^T = K Q^T): lane (q = lane & 31, hi) holds Q[q][16 ks + 8 hi .. + 7] of q-block x
  vec8 qf[2][4];
  int qrow[2];
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    qrow[x] = q0 + 32 * x + lq;
    const unsigned short* Qr = p.Q + b * p.qbs + (long long)qrow[x] * p.qrs + h * p.qhs;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int d = ks * 16 + hi * 8;
      u32x4_t raw = {0u, 0u, 0u, 0u};
      if (qrow[x] < p.Sq && d < p.D) raw = *(const u32x4_t*)(Qr + d);
      __builtin_memcpy(&qf[x][ks], &raw, 16);
    }
  }

  acc_zero();
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};

  // LDS read offsets (attn_kernel's, KCH = 8)
  const int kkey = (lq >> 1) & 7;
  const unsigned k_rd = (unsigned)lq * KROW;
  const int vg = lane >> 4, vR = lane & 15;
  const int v_r0 = 8 * (vg >> 1) + (vR >> 2);
  const int v_key = ((v_r0 >> 1) & 1) << 2;
  const unsigned v_rd = (unsigned)v_r0 * KROW + (unsigned)(vR & 1) * 8u;
  unsigned v_col[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) v_col[dt] = (unsigned)(((4 * dt + 2 * (vg & 1) + ((vR & 3) >> 1)) ^ v_key) << 4);
  vec8 ones8;
  {
    const unsigned one2 = pack2<TT>(1.0f, 1.0f);
    const u32x4_t o4 = {one2, one2, one2, one2};
    __builtin_memcpy(&ones8, &o4, 16);
    asm volatile("" : "+v"(ones8));          // opaque: otherwise re-materialised from SGPRs (2 v_mov_b64) ahead of every row-sum MFMA
  }
  const float c = p.scale_log2;

  auto read_k = [&](const unsigned char* sK, const int m) -> vec8 {      // m = 2 ks + jb
    return *(const vec8*)(sK + (m & 1) * 32 * KROW + k_rd + (((2 * (m >> 1) + hi) ^ kkey) << 4));
  };
  // V^T fragments through inline-asm ds_read_b64_tr_b16: behind the builtin the compiler puts an s_waitcnt vmcnt(0) (it cannot tell the
  // read from the LDS-DMA writes in flight), which would drain the K/V prefetch at every tile. The price: the compiler does not count
  // these reads either, so their consumer carries its own s_waitcnt lgkmcnt(0) (mfma32_agpr<.., true>).
  const unsigned lds0 = (unsigned)(unsigned long long)SX_LDS_PTR(smem);
  const unsigned v_adr[2] = {lds0 + v_rd + v_col[0], lds0 + v_rd + v_col[1]};
  unsigned k_adr[4];                         // K fragment (A operand) of k-step ks: row lane & 31 of a 32-key block, swizzled chunk 2 ks + hi
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) k_adr[ks] = lds0 + k_rd + (unsigned)(((2 * ks + hi) ^ kkey) << 4);
  auto read_k_asm = [&](const unsigned adr, const int off) -> vec8 {
    vec8 kf;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kf) : "v"(adr), "i"(off));
    return kf;
  };
  auto read_v = [&](const unsigned adr, const int off) -> vec8 {    // off: byte offset of the 16-key step (a literal after unrolling)
    u32x2_t lo, hi4;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(adr), "i"(off));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi4) : "v"(adr), "i"(off + 4 * KROW));
    vec8 vf;
    __builtin_memcpy(&vf, &lo, 8);
    __builtin_memcpy((char*)&vf + 8, &hi4, 8);
    return vf;
  };

  f32x16_t sA[2][2], sB[2][2];   // [q-block][32-key block]
  vec8 pb[2][4];                 // P of the current tile, [q-block][16-key step]
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const u32x4_t z4 = {0u, 0u, 0u, 0u};
      __builtin_memcpy(&pb[x][kk], &z4, 16);
    }

  // S(0): plain MFMA cluster
  wait_vmcnt_attn<PRO_DMA - 2>();            // K0 landed
  __builtin_amdgcn_s_barrier();
  {
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int r = 0; r < 16; ++r) sA[x][jb][r] = 0.f;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const vec8 kf = read_k(smem, m);
#pragma unroll
      for (int x = 0; x < 2; ++x) sA[x][m & 1] = TT::mfma32(kf, qf[x][m >> 1], sA[x][m & 1]);
    }
  }
  wait_vmcnt_attn<PRO_DMA - 6>();            // K1, V0 landed
  __builtin_amdgcn_s_barrier();

  unsigned pacc[4] = {0u, 0u, 0u, 0u};
  float mc[2] = {0.f, 0.f};                  // -m_run * c of the two q-blocks
  constexpr float THR = (OPT & 1) ? 8.0f : 0.0f;
  // ---- one KV tile: softmax(t) on s, S(t+1) into sn, O += V(t)^T P(t)^T -----------------------------------------------------
  auto tile = [&](const int kvt, f32x16_t (&s)[2][2], f32x16_t (&sn)[2][2]) {
    // LDS bases of the tile: K(t+1) (S of the next tile), V(t), V(t-1) (carried P·V step); slots are runtime values (8 slots x 16 KB
    // exceed the 16-bit DS offset field), added once per tile to the six lane-constant fragment addresses
    const unsigned oKn = (unsigned)(((kvt + 1) & (NSLOT - 1)) * STAGE), oV = (unsigned)((kvt & (NSLOT - 1)) * STAGE + K_BYTES);
    const unsigned oVp = (unsigned)(((kvt + NSLOT - 1) & (NSLOT - 1)) * STAGE + K_BYTES);
    unsigned kA[4], vA[2], vP[2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kA[ks] = k_adr[ks] + oKn;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) { vA[dt] = v_adr[dt] + oV; vP[dt] = v_adr[dt] + oVp; }
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
    if (PROBE) t0 = __builtin_amdgcn_s_memtime();
    if (!AB_DMA) {
      stage_k((kvt + NSLOT - 1) & (NSLOT - 1), kvt + NSLOT - 1);
      stage_v((kvt + NSLOT - 2) & (NSLOT - 1), kvt + NSLOT - 2);
    }
    if (PROBE) { __builtin_amdgcn_sched_barrier(0); t1 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
    const int kv0 = kvt * 64;
    if (kv0 + 64 > p.Skv) {                  // wave-uniform: the last tile of a ragged Skv
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int kv = kv0 + 32 * jb + (r & 3) + 4 * ((r >> 2) & 1) + 8 * hi + 16 * (r >> 3);
            if (kv >= p.Skv) s[x][jb][r] = -INFINITY;
          }
    }
    // MFMA queue of the tile, in pairs (q-block a then b on one fragment):
    //   CARRY: g 0..2 = P·V step 3 of tile t-1 (ones | d tile 0 | d tile 1), g 3..10 = S(t+1) m = 0..7, then P·V steps 0..2 of tile t
    //   else:  g 0..7 = S(t+1), then P·V steps 0..3
    constexpr int PPV = ONES ? 3 : 2;                              // pairs per P·V step
    constexpr int G_S0 = CARRY ? PPV : 0, G_PV0 = G_S0 + 8, G_END = G_PV0 + PPV * (CARRY ? 3 : 4);
    // LDS fragments of the tile in consumption order, f = 0..15: CARRY: V(t-1) step 3 d tiles 0, 1 | K(t+1) m = 0..7 | V(t) steps 0..2
    // (d tile 0, 1 each); else K(t+1) m = 0..7 | V(t) steps 0..3. All through inline asm (a K fragment = 1 ds_read_b128, a V^T
    // fragment = 2 ds_read_b64_tr_b16), FD fragments ahead of their MFMA pair into a ring of FD + 1 registers sets; the consumer waits
    // with a COUNTED lgkmcnt = the reads issued behind its fragment (LDS returns in order). One wave per SIMD has nobody to hide an
    // exposed LDS round trip behind: with one fragment of look-ahead every pair stalled on it.
    constexpr int FD = 3, NFRAG = 16, F_K0 = CARRY ? 2 : 0, F_V0 = F_K0 + 8;
    vec8 fr[FD + 1];
    auto frag_ops = [&](const int f) -> int { return (f >= F_K0 && f < F_V0) ? 1 : 2; };
    auto issue_frag = [&](const int f) {
      if (f >= NFRAG) return;
      if (f < F_K0) {
        fr[f & FD] = read_v(vP[f], 3 * 16 * KROW);
      } else if (f < F_V0) {
        const int m = f - F_K0;                                    // m = 2 ks + jb
        fr[f & FD] = read_k_asm(kA[m >> 1], (m & 1) * 32 * KROW);
      } else {
        fr[f & FD] = read_v(vA[(f - F_V0) & 1], ((f - F_V0) >> 1) * 16 * KROW);
      }
    };
    auto wait_frag = [&](const int f) {                             // fragment f has landed; the wait is tied to its registers
      int n = 0;
#pragma unroll
      for (int i = f + 1; i < f + FD && i < NFRAG; ++i) n += frag_ops(i);
      asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(fr[f & FD]) : "i"(n));
    };
    auto mf = [&](const int j) {                                    // MFMA j of the queue (pair j >> 1, q-block j & 1)
      const int g = j >> 1, x = j & 1;
      if (g >= G_END) return;
      int f = -1;                                                   // the pair's fragment (none: a row-sum pair)
      if (g < G_S0) {
        const int sub = g - (ONES ? 1 : 0);
        if (sub >= 0) f = sub;
        if (x == 0 && f >= 0) wait_frag(f);
        if (AB_PV) { asm volatile("" ::"v"(fr[f < 0 ? 0 : f & FD]), "v"(pb[x][3])); }
        else if (sub < 0) acc_mfma<TT>(3 * x + 2, ones8, pb[x][3]);
        else acc_mfma<TT>(3 * x + sub, fr[f & FD], pb[x][3]);
      } else if (g < G_PV0) {
        const int m = g - G_S0;
        f = F_K0 + m;
        if (x == 0) wait_frag(f);
        if (AB_S) {
          asm volatile("" : "+v"(sn[x][m & 1]) : "v"(fr[f & FD]));
        } else if (m < 2) {
          f32x16_t z;
#pragma unroll
          for (int r = 0; r < 16; ++r) z[r] = 0.f;
          sn[x][m & 1] = TT::mfma32(fr[f & FD], qf[x][0], z);
        } else {
          sn[x][m & 1] = TT::mfma32(fr[f & FD], qf[x][m >> 1], sn[x][m & 1]);
        }
      } else {
        const int kk = (g - G_PV0) / PPV, sub = (g - G_PV0) % PPV - (ONES ? 1 : 0);
        if (sub >= 0) f = F_V0 + 2 * kk + sub;
        if (x == 0 && f >= 0) wait_frag(f);
        if (AB_PV) { asm volatile("" ::"v"(fr[f < 0 ? 0 : f & FD]), "v"(pb[x][kk])); }
        else if (sub < 0) acc_mfma<TT>(3 * x + 2, ones8, pb[x][kk]);
        else acc_mfma<TT>(3 * x + sub, fr[f & FD], pb[x][kk]);
      }
      if (x == 0 && f >= 0) issue_frag(f + FD);                     // behind the pair's first MFMA
    };

    // ---- block 1: row maxima of both q-blocks, running-max update; MFMAs: the carried P·V pairs + the first S pairs ----------
    float mloc[2] = {-INFINITY, -INFINITY}, mloc2[2] = {-INFINITY, -INFINITY}, mrow[2];
    bool need[2];
    constexpr int N1 = CARRY ? 2 * PPV + 6 : 10;                    // MFMAs issued in block 1 (9 VALU pieces)
    int j = 0;
#pragma unroll
    for (int f = 0; f < FD; ++f) issue_frag(f);
    if (PRIO) __builtin_amdgcn_s_setprio(1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 9   // explicit counts: a plain "#pragma unroll" gives up silently above -pragma-unroll-threshold, and j must fold
    for (int pc = 0; pc < 9; ++pc) {
      if (pc < 8) {
        // four independent max3 chains per piece (both q-blocks x even / odd S^T rows): no issue waits on the previous one
        if (!AB_SM)
#pragma unroll
          for (int x = 0; x < 2; ++x) {
            mloc[x] = max3f(mloc[x], s[x][0][2 * pc], s[x][1][2 * pc]);
            mloc2[x] = max3f(mloc2[x], s[x][0][2 * pc + 1], s[x][1][2 * pc + 1]);
          }
      } else {
        float xa0 = fmaxf(mloc[0], mloc2[0]), xb0 = xa0, xa1 = fmaxf(mloc[1], mloc2[1]), xb1 = xa1;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\ts_nop 1"
                     : "+v"(xa0), "+v"(xb0), "+v"(xa1), "+v"(xb1));
        mrow[0] = fmaxf(xa0, xb0);
        mrow[1] = fmaxf(xa1, xb1);
#pragma unroll
        for (int x = 0; x < 2; ++x) need[x] = (mrow[x] - m_run[x]) * c > THR;   // -inf reference (first tile): +inf > THR; all-masked row: NaN > THR is false
      }
      __builtin_amdgcn_sched_barrier(0);
      static_assert(N1 / 9 == 1, "one MFMA per piece, a second one behind the first N1 - 9 pieces");
      mf(j++);
      if (pc < N1 % 9) mf(j++);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (PRIO) __builtin_amdgcn_s_setprio(0);
    // Reference-max update (T13 "defer-max", cdna_hip_programming.md): the exponent reference m_run of a row moves only when the tile's
    // maximum exceeds it by more than THR (log2 units), so P stays <= 2^THR instead of <= 1 — the same RELATIVE rounding in fp16 /
    // bf16, row sums and O in fp32 — and the O rescale (3 x 16 accumulator registers per q-block through v_accvgpr_read / write)
    // leaves the per-tile path. With THR = 0 (exact running max) some row of a 64-row wave raises its maximum in almost every tile:
    // measured 1500 of 3860 cycles per tile.
    if (__builtin_amdgcn_ballot_w64(need[0] || need[1]) != 0) {   // wave-uniform
#pragma unroll
      for (int x = 0; x < 2; ++x)
        if (__builtin_amdgcn_ballot_w64(need[x]) != 0) {
          const float m_new = fmaxf(m_run[x], mrow[x]);
          const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
          const float al = (m_run[x] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((m_run[x] - m_safe) * c);
          m_run[x] = m_new;
          mc[x] = -m_safe * c;
#pragma unroll
          for (int i = 0; i < NDT; ++i) acc_scale(3 * x + i, al);
        }
    }
    // ---- block 2: 32 quarter-chunks (2 scores of one q-block each), ONE MFMA of the queue behind each. Software-pipelined so that no
    // instruction issues right behind the one it depends on: piece q = { 2 v_exp of chunk q | 2 v_fma of chunk q + 1 | pack of chunk q }
    unsigned pw[2][4][4];
    float psum[2] = {0.f, 0.f};
    auto scale_shift = [&](const int qc, float& e0, float& e1) {   // e = s * c - m * c of the chunk's two scores
      const int kk = qc >> 3, x = (qc >> 2) & 1, w = qc & 3, jb = kk >> 1, s2 = kk & 1;
      if (SCALAR) {
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(e0) : "v"(s[x][jb][8 * s2 + 2 * w]), "v"(c), "v"(mc[x]));
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(e1) : "v"(s[x][jb][8 * s2 + 2 * w + 1]), "v"(c), "v"(mc[x]));
      } else {
        typedef float f2_t __attribute__((ext_vector_type(2)));
        const f2_t sv = {s[x][jb][8 * s2 + 2 * w], s[x][jb][8 * s2 + 2 * w + 1]};
        const f2_t e = __builtin_elementwise_fma(sv, (f2_t){c, c}, (f2_t){mc[x], mc[x]});
        e0 = e[0]; e1 = e[1];
      }
    };
    float ea[2], eb[2];                                              // chunk q in ea (q even) / eb (q odd)
    if (!AB_SM) scale_shift(0, ea[0], ea[1]);
    if (PRIO) __builtin_amdgcn_s_setprio(1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 32
    for (int qc = 0; qc < 32; ++qc) {
      const int kk = qc >> 3, x = (qc >> 2) & 1, w = qc & 3;
      if (!AB_SM) {
        float (&ec)[2] = (qc & 1) ? eb : ea;
        float (&en)[2] = (qc & 1) ? ea : eb;
        const float p0 = AB_EXP ? ec[0] : __builtin_amdgcn_exp2f(ec[0]);
        const float p1 = AB_EXP ? ec[1] : __builtin_amdgcn_exp2f(ec[1]);
        if (qc + 1 < 32) scale_shift(qc + 1, en[0], en[1]);
        pw[x][kk][w] = pack2<TT>(p0, p1);
        if (!ONES) psum[x] = TT::pair_sum(pw[x][kk][w], psum[x]);
      }
      if (w == 3 && !AB_SM) __builtin_memcpy(&pb[x][kk], pw[x][kk], 16);
      __builtin_amdgcn_sched_barrier(0);
      // availability: P·V step kk needs the quarter-chunks 8 kk .. 8 kk + 7 (both q-blocks); S pairs are always available
      {
        const int g = j >> 1;
        bool okj = g < G_END;
        if (okj && g >= G_PV0) okj = qc >= 8 * ((g - G_PV0) / PPV) + 7;
        if (okj) mf(j++);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // whatever the availability rule held back (without CARRY: the last P·V step)
#pragma unroll
    for (int k2 = 0; k2 < 2 * PPV * 2; ++k2)
      if ((j >> 1) < G_END) mf(j++);
    if (PRIO) __builtin_amdgcn_s_setprio(0);
    if (!ONES) { l_run[0] += psum[0]; l_run[1] += psum[1]; }
    __builtin_amdgcn_sched_barrier(0);
    if (PROBE) { t2 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
    wait_vmcnt_attn<AB_DMA ? 0 : 4 * (NSLOT - 3)>();   // everything but the five youngest K/V pairs: K(t+2), V(t+1) have landed
    if (PROBE) { t3 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (PROBE) {
      t4 = __builtin_amdgcn_s_memtime();
      pacc[0] += (unsigned)(t1 - t0); pacc[1] += (unsigned)(t2 - t1); pacc[2] += (unsigned)(t3 - t2); pacc[3] += (unsigned)(t4 - t3);
    }
  };

  for (int kvt = 0; kvt < nt; kvt += 2) {
    tile(kvt, sA, sB);
    if (kvt + 1 >= nt) break;
    tile(kvt + 1, sB, sA);
  }
  wait_vmcnt_attn<0>();                       // zero-fill DMAs of the tiles past the end must not outlive the LDS allocation
  if (CARRY) {                                // P·V step 3 of the last tile
    const unsigned sl = (unsigned)(((nt - 1) & (NSLOT - 1)) * STAGE);
#pragma unroll
    for (int x = 0; x < 2; ++x)
      acc_mfma<TT>(3 * x + 2, ones8, pb[x][3]);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      vec8 vf = read_v(v_adr[dt] + sl, K_BYTES + 3 * 16 * KROW);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vf));
      acc_mfma<TT>(dt, vf, pb[0][3]);
      acc_mfma<TT>(3 + dt, vf, pb[1][3]);
    }
  }
  if (PROBE && blockIdx.x == 8 && lane == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) ((unsigned*)p.O)[wave * 4 + i] = pacc[i];
    return;
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // asm MFMA results → accumulator reads (the compiler cannot see into the asm)

  // ---- epilogue: O[q][d] = o / l -----------------------------------------------------------------------------------------
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    float ls[16], ov[2][16];
    acc_read(3 * x + 2, ls);
    acc_read(3 * x, ov[0]);
    acc_read(3 * x + 1, ov[1]);
    const float l_tot = ls[0];
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (qrow[x] < p.Sq) {
      unsigned short* Or = p.O + b * p.obs + (long long)qrow[x] * p.ors + (long long)h * p.D;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = 32 * dt + 8 * g + 4 * hi;
          if (d < p.D) {
            u32x2_t w;
            w[0] = pack2<TT>(ov[dt][4 * g] * inv, ov[dt][4 * g + 1] * inv);
            w[1] = pack2<TT>(ov[dt][4 * g + 2] * inv, ov[dt][4 * g + 3] * inv);
            *(u32x2_t*)(Or + d) = w;
          }
        }
    }
  }
#endif
}

// ---- small generic attention: one wave per (b, h, q) row; scores staged in LDS ------------------------------
struct AttnSmallP {
  const unsigned short* Q;
  const unsigned short* K;
  const unsigned short* V;
  unsigned short* O;
  int B, H, Sq, Skv, D;
  long long qbs, qrs, qhs, kbs, krs, khs, vbs, vrs, vhs, obs, ors;
  float scale;
};
constexpr int SMALL_MAX_KV = 1536;
constexpr int SMALL_MAX_D = 256;

template <typename TT>
__global__ __launch_bounds__(256) void attn_small_kernel(const AttnSmallP p) {
  __shared__ float sc[4][SMALL_MAX_KV];
  __shared__ float qs[4][SMALL_MAX_D];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long long row = (long long)blockIdx.x * 4 + w;
  const long long total = (long long)p.B * p.H * p.Sq;
  if (row >= total) return;
  const int q = (int)(row % p.Sq);
  const int h = (int)((row / p.Sq) % p.H);
  const int b = (int)(row / ((long long)p.Sq * p.H));
  const unsigned short* Qr = p.Q + b * p.qbs + (long long)q * p.qrs + h * p.qhs;
  for (int d = lane; d < p.D; d += 64) qs[w][d] = TT::to_f32(Qr[d]) * p.scale;
  __builtin_amdgcn_wave_barrier();
  const unsigned short* Kb = p.K + b * p.kbs + h * p.khs;
  float mx = -INFINITY;
  for (int kv = lane; kv < p.Skv; kv += 64) {
    const unsigned short* Kr = Kb + (long long)kv * p.krs;
    float acc = 0.f;
    for (int d = 0; d < p.D; d += 8) {
      const u32x4_t raw = *(const u32x4_t*)(Kr + d);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc += TT::to_f32(raw[e] & 0xffff) * qs[w][d + 2 * e];
        acc += TT::to_f32(raw[e] >> 16) * qs[w][d + 2 * e + 1];
      }
    }
    sc[w][kv] = acc;
    mx = fmaxf(mx, acc);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int kv = lane; kv < p.Skv; kv += 64) {
    const float e = __expf(sc[w][kv] - mx);
    sc[w][kv] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  __builtin_amdgcn_wave_barrier();
  const float inv = 1.0f / sum;
  const unsigned short* Vb = p.V + b * p.vbs + h * p.vhs;
  unsigned short* Or = p.O + b * p.obs + (long long)q * p.ors + (long long)h * p.D;
  for (int d2 = lane; d2 < p.D / 2; d2 += 64) {
    float a0 = 0.f, a1 = 0.f;
    for (int kv = 0; kv < p.Skv; ++kv) {
      const unsigned v = *(const unsigned*)(Vb + (long long)kv * p.vrs + 2 * d2);
      const float pr = sc[w][kv];
      a0 += pr * TT::to_f32(v & 0xffff);
      a1 += pr * TT::to_f32(v >> 16);
    }
    *(unsigned*)(Or + 2 * d2) = pack2<TT>(a0 * inv, a1 * inv);
  }
}

}  // namespace sxk_attn
using namespace sxk_attn;

#ifndef SX_ATTN_OPT_D64
#define SX_ATTN_OPT_D64 34     // OPT bits of the shipped head_dim-64 kernel (see attn_kernel; round-4 lab: profiles/r4_attn_lab_opt_variants.log)
#endif
static int g_attn_variant = 0;   // tuning hook (tools/lab/attn_lab): 0 = shipped kernel; 16 + OPT = the head_dim-64 kernel with those OPT bits
static int g_attn_lds_pad = 0;   // lab only: extra dynamic LDS (KiB) per workgroup of the head_dim-64 variants, to cap the workgroups per CU
extern "C" int sx_attention_variant(int v) {
  g_attn_lds_pad = v >= 100000 ? (v - 100000) / 1000 : 0;       // 100000 + 1000 KiB + variant
  g_attn_variant = v >= 100000 ? v % 1000 : v;
  return SX_OK;
}

extern "C" int sx_attention(const sx_attn_args* a, void* stream) {
  SX_CHECK(a && a->Q && a->K && a->V && a->O, "sx_attention: null pointer");
  SX_CHECK(a->dtype == SX_F16 || a->dtype == SX_BF16, "sx_attention: dtype");
  SX_CHECK(a->D % 8 == 0 && a->D >= 8 && a->D <= 128, "sx_attention: head_dim %d unsupported", a->D);
  SX_CHECK(a->Sq > 0 && a->Skv > 0 && a->B > 0 && a->H > 0, "sx_attention: empty problem");
  SX_CHECK(a->q_row_stride % 8 == 0 && a->q_head_stride % 8 == 0 && a->q_batch_stride % 8 == 0 &&
               a->k_row_stride % 8 == 0 && a->k_head_stride % 8 == 0 && a->k_batch_stride % 8 == 0 &&
               a->v_row_stride % 8 == 0 && a->v_head_stride % 8 == 0 && a->v_batch_stride % 8 == 0,
           "sx_attention: Q/K/V strides must be multiples of 8 elements (16 B)");
  SX_CHECK(a->o_row_stride % 4 == 0 && a->o_batch_stride % 4 == 0, "sx_attention: O strides");
  SX_CHECK(!a->causal || a->Skv >= a->Sq, "sx_attention: causal needs Skv >= Sq");
  SX_CHECK(((int64_t)(a->Skv - 1) * a->k_row_stride + a->D) * 2 < 0x7fffffffll, "sx_attention: K range too large");
  SX_CHECK(((int64_t)(a->Skv - 1) * a->v_row_stride + a->D) * 2 < 0x7fffffffll, "sx_attention: V range too large");
  AttnP p;
  p.Q = (const unsigned short*)a->Q; p.K = (const unsigned short*)a->K; p.V = (const unsigned short*)a->V;
  p.O = (unsigned short*)a->O;
  p.B = a->B; p.H = a->H; p.Sq = a->Sq; p.Skv = a->Skv; p.D = a->D;
  p.q_tiles = (a->Sq + 127) / 128;
  p.causal = a->causal;
  p.qbs = a->q_batch_stride; p.qrs = a->q_row_stride; p.qhs = a->q_head_stride;
  p.kbs = a->k_batch_stride; p.krs = a->k_row_stride; p.khs = a->k_head_stride;
  p.vbs = a->v_batch_stride; p.vrs = a->v_row_stride; p.vhs = a->v_head_stride;
  p.obs = a->o_batch_stride; p.ors = a->o_row_stride;
  p.scale_log2 = a->scale * 1.4426950408889634f;
  const int grid = a->B * a->H * p.q_tiles;
  hipStream_t st = (hipStream_t)stream;
  const int dp = a->D <= 64 ? 64 : 128;
  const size_t lds = 2 * (size_t)(2 * 64 * dp * 2);     // 2 stages x (K tile + V tile)
  if (g_attn_variant >= 16 && g_attn_variant < 16 + 64 && dp == 64) {   // A/B: OPT bits of attn_kernel (tools/lab/attn_lab)
    const bool bf = a->dtype == SX_BF16;
    const size_t ldsp = lds + (size_t)g_attn_lds_pad * 1024;
#define ATTN_OPT_CASE(O) case O: \
      if (ldsp > 65536) (void)hipFuncSetAttribute(bf ? (const void*)attn_kernel<BF16, 64, O> : (const void*)attn_kernel<F16, 64, O>, \
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp);                          \
      if (bf) hipLaunchKernelGGL((attn_kernel<BF16, 64, O>), dim3(grid), dim3(256), ldsp, st, p);                                 \
      else hipLaunchKernelGGL((attn_kernel<F16, 64, O>), dim3(grid), dim3(256), ldsp, st, p);                                     \
      break;
    switch (g_attn_variant - 16) {
      ATTN_OPT_CASE(1) ATTN_OPT_CASE(2) ATTN_OPT_CASE(3) ATTN_OPT_CASE(4) ATTN_OPT_CASE(7) ATTN_OPT_CASE(8) ATTN_OPT_CASE(9)
      ATTN_OPT_CASE(10) ATTN_OPT_CASE(11) ATTN_OPT_CASE(16) ATTN_OPT_CASE(17) ATTN_OPT_CASE(18) ATTN_OPT_CASE(19)
      ATTN_OPT_CASE(34) ATTN_OPT_CASE(35) ATTN_OPT_CASE(50)
      default: SX_CHECK(false, "sx_attention: variant %d not built", g_attn_variant);
    }
#undef ATTN_OPT_CASE
    SX_HIP_LAUNCH_CHECK();
    return SX_OK;
  }
  if (g_attn_variant >= 2000 && g_attn_variant < 2000 + 1024 && dp == 64 && !a->causal) {   // A/B: the one-wave-per-SIMD head_dim-64 kernel
    const bool bf = a->dtype == SX_BF16;
    p.q_tiles = (a->Sq + 255) / 256;
    const int grid64 = a->B * a->H * p.q_tiles;
    const size_t lds64 = 8 * 16384;
#define ATTN64_CASE(O) case O:                                                                                                 \
      (void)hipFuncSetAttribute(bf ? (const void*)attn64_kernel<BF16, O> : (const void*)attn64_kernel<F16, O>,                      \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds64);                                          \
      if (bf) hipLaunchKernelGGL((attn64_kernel<BF16, O>), dim3(grid64), dim3(256), lds64, st, p);                                \
      else hipLaunchKernelGGL((attn64_kernel<F16, O>), dim3(grid64), dim3(256), lds64, st, p);                                    \
      break;
    switch (g_attn_variant - 2000) {
      ATTN64_CASE(6) ATTN64_CASE(7) ATTN64_CASE(5) ATTN64_CASE(3) ATTN64_CASE(23) ATTN64_CASE(39) ATTN64_CASE(71) ATTN64_CASE(103) ATTN64_CASE(135) ATTN64_CASE(263) ATTN64_CASE(519)
      default: SX_CHECK(false, "sx_attention: variant %d not built", g_attn_variant);
    }
#undef ATTN64_CASE
    SX_HIP_LAUNCH_CHECK();
    return SX_OK;
  }
  if (a->dtype == SX_BF16) {
    if (dp == 64) hipLaunchKernelGGL((attn_kernel<BF16, 64, SX_ATTN_OPT_D64>), dim3(grid), dim3(256), lds, st, p);
    else hipLaunchKernelGGL((attn_kernel<BF16, 128>), dim3(grid), dim3(256), lds, st, p);
  } else {
    if (dp == 64) hipLaunchKernelGGL((attn_kernel<F16, 64, SX_ATTN_OPT_D64>), dim3(grid), dim3(256), lds, st, p);
    else hipLaunchKernelGGL((attn_kernel<F16, 128>), dim3(grid), dim3(256), lds, st, p);
  }
  SX_HIP_LAUNCH_CHECK();
  return SX_OK;
}

extern "C" int sx_attention_small(const sx_attn_small_args* a, void* stream) {
  SX_CHECK(a && a->Q && a->K && a->V && a->O, "sx_attention_small: null pointer");
  SX_CHECK(a->dtype == SX_F16 || a->dtype == SX_BF16, "sx_attention_small: dtype");
  SX_CHECK(a->D % 8 == 0 && a->D <= SMALL_MAX_D, "sx_attention_small: head_dim %d", a->D);
  SX_CHECK(a->Skv > 0 && a->Skv <= SMALL_MAX_KV, "sx_attention_small: Skv=%d exceeds %d", a->Skv, SMALL_MAX_KV);
  SX_CHECK(a->k_row_stride % 8 == 0 && a->k_head_stride % 8 == 0 && a->k_batch_stride % 8 == 0,
           "sx_attention_small: K strides must be multiples of 8");
  SX_CHECK(a->v_row_stride % 2 == 0 && a->v_head_stride % 2 == 0 && a->o_row_stride % 2 == 0,
           "sx_attention_small: V/O strides must be even");
  AttnSmallP p;
  p.Q = (const unsigned short*)a->Q; p.K = (const unsigned short*)a->K; p.V = (const unsigned short*)a->V;
  p.O = (unsigned short*)a->O;
  p.B = a->B; p.H = a->H; p.Sq = a->Sq; p.Skv = a->Skv; p.D = a->D;
  p.qbs = a->q_batch_stride; p.qrs = a->q_row_stride; p.qhs = a->q_head_stride;
  p.kbs = a->k_batch_stride; p.krs = a->k_row_stride; p.khs = a->k_head_stride;
  p.vbs = a->v_batch_stride; p.vrs = a->v_row_stride; p.vhs = a->v_head_stride;
  p.obs = a->o_batch_stride; p.ors = a->o_row_stride;
  p.scale = a->scale;
  const long long rows = (long long)a->B * a->H * a->Sq;
  const dim3 grid((unsigned)((rows + 3) / 4));
  if (a->dtype == SX_BF16) hipLaunchKernelGGL(attn_small_kernel<BF16>, grid, dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(attn_small_kernel<F16>, grid, dim3(256), 0, (hipStream_t)stream, p);
  SX_HIP_LAUNCH_CHECK();
  return SX_OK;
}
