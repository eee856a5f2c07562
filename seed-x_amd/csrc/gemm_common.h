// Shared between the two GEMM translation units (gemm.hip: lock-step tiles + dispatch, gemm_pp.hip: ping-pong 256-row tiles).
#pragma once
#include "sx_common.h"

namespace sxk_gemm {

struct GemmP {
  const void* A;
  const void* W;
  void* C;
  const float* bias;
  const float* bias2d;
  const float* residual;
  int M, N, K, ldc, ldr, n_valid, res_mod, bias2d_rows, out_dtype, act, glu;
  int Kw;                         // row length of W in elements: = K, or K / 2 with sx_gemm_args.a_planes = 2 (A = [hi | lo]: the k loop walks W twice)
  int Hin, Win, Cin, Hout, Wout, stride, upsample, ldb2, pad;
  int tiles_m, tiles_n, xm, xn;   // tile grid and its XCD partition (xm x xn == 8, or 0 = linear remap)
  int gm;                         // tile-rows per group of the in-XCD traversal
  int res_init;                   // residual is the accumulators' initial value (act == none, no GLU): no epilogue loads
  int n_tiles;                    // persistent kernels: logical grid size (blocks loop bid += gridDim.x)
  int tune;                       // A/B switches (sx_gemm_force_tile 600 + mask): bit 0 = GLU epilogue keeps its 8-byte stores
  unsigned a_bytes, w_bytes;
  unsigned long long* dbg;        // tuning hook: per-block s_memtime stamps [block][4] = start, first tile landed, main loop done, end
  // fused GroupNorm statistics of the stored fp32 output (sx_gemm_gn; ping-pong tiles only): stats[sample][group][2] += (sum, sum of
  // squares) over the tile's rows and the group's channels; sample = row / gn_rows, group = column / gn_cpg
  double* gn_stats;
  int gn_cpg, gn_groups, gn_rows;
  // LayerNorm folded into the GEMMs either side of it (sx_gemm_ln; ping-pong tiles only).
  //   producer (fp32 output): also store the output rounded to the operand dtype (ln_x16 [M][ln_ldx]) and add this tile's
  //     per-row (sum, sum of squares) to ln_out[M][2] (fp64 atomics);
  //   consumer (A = that 16-bit copy, W = weight with gamma folded in): out = rstd_m (acc - mu_m cs_n) + bias'_n, with (mu, rstd)
  //     of row m from ln_in[M][2], cs_n = sum_k W'[n][k]
  void* ln_x16;
  double* ln_out;
  const double* ln_in;
  const float* ln_cs;
  int ln_ldx;
  float ln_eps, ln_inv_dim;
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Tile → XCD mapping. Block b is observed on XCD b % 8 (performance heuristic only). Each XCD gets a compact
// (tiles_m/xm) x (tiles_n/xn) rectangle of the tile grid, so its private L2 holds A-panel/xm + W-panel/xn instead of
// re-streaming whole panels from Infinity Cache / HBM (PMC: profiles/r1_pmc_hbm.json). Returns false for the padding
// blocks of an uneven split.
__device__ __forceinline__ bool tile_coords(const GemmP& p, int bid, int nblk, int& tile_m, int& tile_n) {
  if (p.xm == 0) {
    const int t = xcd_remap(bid, nblk);
    tile_m = t % p.tiles_m;
    tile_n = t / p.tiles_m;
    return true;
  }
  const int xcd = bid & 7, idx = bid >> 3;
  const int xr = xcd / p.xn, xc = xcd - xr * p.xn;
  const int ms = xr * p.tiles_m / p.xm, me = (xr + 1) * p.tiles_m / p.xm;
  const int ns = xc * p.tiles_n / p.xn, ne = (xc + 1) * p.tiles_n / p.xn;
  const int rm = me - ms, rn = ne - ns;
  if (idx >= rm * rn) return false;
  // grouped order inside the rectangle: the ~32 blocks resident on an XCD at one time form a gm x (32/gm) patch (not a
  // 32 x 1 column), and consecutive rounds keep the same gm A tile-rows while sweeping n → per round the XCD's L2 pulls
  // gm + 32/gm operand tile-rows instead of 33 (PMC: FETCH_SIZE of the GEGLU GEMM 8.8x → see profiles/r1_pmc_summary.json)
  const int per_group = p.gm * rn;
  const int g = idx / per_group, first = g * p.gm;
  const int gsz = (rm - first) < p.gm ? (rm - first) : p.gm;
  const int r = idx - g * per_group;
  tile_m = ms + first + r % gsz;
  tile_n = ns + r / gsz;
  return true;
}

// host side: fill tiles_m/tiles_n/xm/xn/gm for a BM x BN tiling; returns the launch grid
inline int plan_grid(GemmP& p, int BM, int BN, int xcd_2d, int gm_force) {
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  int grid = p.tiles_m * p.tiles_n;
  p.xm = p.xn = 0;
  p.gm = 1;
  if (grid >= 16 && xcd_2d) {
    // choose the 8-way split that minimises fabric traffic  A_bytes * xn + W_bytes * xm  among the least padded ones
    const double ab = (double)p.M * p.K, wb = (double)p.N * p.K;
    double best = 1e300;
    for (int xm = 1; xm <= 8; xm *= 2) {
      const int xn = 8 / xm;
      if (xm > p.tiles_m || xn > p.tiles_n) continue;
      const long padded = 8L * ((p.tiles_m + xm - 1) / xm) * ((p.tiles_n + xn - 1) / xn);
      const double cost = (ab * xn + wb * xm) * (1.0 + 4.0 * (double)(padded - grid) / grid);
      if (cost < best) { best = cost; p.xm = xm; p.xn = xn; }
    }
    if (p.xm) grid = 8 * ((p.tiles_m + p.xm - 1) / p.xm) * ((p.tiles_n + p.xn - 1) / p.xn);
    p.gm = gm_force > 0 ? gm_force : 8;   // tools/bench_gm.py: 8 is best or within 1 % on every multi-round shape
  }
  p.n_tiles = grid;
  return grid;
}

// ping-pong tiles (gemm_pp.hip). bn = 256 | 320. Returns SX_OK, or -1 when the epilogue combination is not instantiated
// (caller falls back to the lock-step kernel).
int launch_pp(const GemmP& p, int dtype, int bn, int a_mode, hipStream_t st);
bool pp_supported(const GemmP& p, int dtype, int bn, int a_mode);

// persistent strip kernel of the fp32-residual LayerNorm producers (gemm_strip.hip): whole 128-row strips per workgroup, two
// accumulator sets, residual loads / output stores under the main loop. strip_supported: the launch fits its contract.
int launch_strip(const GemmP& p, int dtype, hipStream_t st);
bool strip_supported(const GemmP& p, int a_mode);

extern unsigned long long* g_dbg;
extern int g_gm;
extern int g_xcd_2d;
extern int g_pp_variant;   // tuning hook (sx_gemm_force_tile 4xx)
extern int g_tune;         // tuning hook (sx_gemm_force_tile 600 + mask) → GemmP::tune

}  // namespace sxk_gemm
