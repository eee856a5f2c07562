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

namespace sxk_attn {


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
  float m_run = -INFINITY, l_run = 0.f;

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
    const float m_new = fmaxf(m_run, mloc);
    const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((m_run - m_safe) * c);
    m_run = m_new;
    const float mc = -m_safe * c;
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
#define SX_ATTN_OPT_D64 2      // OPT bits of the shipped head_dim-64 kernel (see attn_kernel; round-4 lab: profiles/r4_attn_lab_opt_variants.log)
#endif
static int g_attn_variant = 0;   // tuning hook (tools/lab/attn_lab): 0 = shipped kernel; 16 + OPT = the head_dim-64 kernel with those OPT bits
extern "C" int sx_attention_variant(int v) { g_attn_variant = v; return SX_OK; }

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
  if (g_attn_variant >= 16 && g_attn_variant < 48 && dp == 64) {   // A/B: OPT bits of attn_kernel (tools/lab/attn_lab)
    const bool bf = a->dtype == SX_BF16;
#define ATTN_OPT_CASE(O) case O: if (bf) hipLaunchKernelGGL((attn_kernel<BF16, 64, O>), dim3(grid), dim3(256), lds, st, p); \
                                 else hipLaunchKernelGGL((attn_kernel<F16, 64, O>), dim3(grid), dim3(256), lds, st, p); break;
    switch (g_attn_variant - 16) {
      ATTN_OPT_CASE(1) ATTN_OPT_CASE(2) ATTN_OPT_CASE(3) ATTN_OPT_CASE(4) ATTN_OPT_CASE(7) ATTN_OPT_CASE(8) ATTN_OPT_CASE(9)
      ATTN_OPT_CASE(10) ATTN_OPT_CASE(11) ATTN_OPT_CASE(16) ATTN_OPT_CASE(17) ATTN_OPT_CASE(18) ATTN_OPT_CASE(19)
      default: SX_CHECK(false, "sx_attention: variant %d not built", g_attn_variant);
    }
#undef ATTN_OPT_CASE
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
