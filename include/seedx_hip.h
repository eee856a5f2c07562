/*
 * seedx_hip.h — C-ABI of libseedx_hip.so: the hand-written gfx950 (MI355X / CDNA4) kernels behind the
 * SEED-X inference hot path (ViT visual encoder → Llama-style LLM prefill/decode → SDXL-adapter UNet loop).
 *
 * The reference (AILab-CVC/SEED-X) has NO native layer: every op below replaces a stock
 * PyTorch / xformers / diffusers call made from the reference's Python modules. Each entry cites the
 * reference call site (file:line under the reference tree) it stands in for. The Python host in
 * `seed-x_amd/` binds these with ctypes (INTEGRATION.md shows the stub a reference maintainer would add).
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless a name ends in `_host`
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); every call only ENQUEUES work
 *   - return value: 0 (SX_OK) on success; non-zero on error, message via sx_last_error()
 *   - 16-bit activations/weights are fp16 or bf16 selected by `dtype`; accumulation is always fp32
 *   - row-major everywhere; weights are [N][K] (torch nn.Linear layout), conv weights [Cout][3][3][Cin]
 */
#ifndef SEEDX_HIP_H
#define SEEDX_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SX_OK 0
#define SX_ERR_INVALID 1
#define SX_ERR_HIP 2

/* element types */
#define SX_F16 0
#define SX_BF16 1
#define SX_F32 2
#define SX_BF16X3 3 /* sx_groupnorm* outputs only: bf16 planes [hi | hi | lo] per row, 3*C columns (see sx_split_bf16) */
#define SX_F16X2 4  /* sx_groupnorm* outputs only: fp16 planes [hi | lo] per row, 2*C columns (sx_split16's row layout): the A operand
                     * of the fp32-grade VAE convs whose weights are exact in fp16 (W duplicated per tap: A.W = Ah.W + Al.W) */
/* OR-ed into a 16-bit OUTPUT dtype of the decode-step producers (sx_layernorm with rows <= 32, sx_attn_decode_b, sx_gemv):
 * the [rows <= 32][cols] result is written as MFMA operand tiles [rows/16][cols/32][16][32] — what sx_gemv reads with x_layout = 1
 * (tile t = columns 32t .. 32t+31 of all 16 rows, 1 KB contiguous; rows >= `rows` of a tile are not written). */
#define SX_TILED16 0x100

/* epilogue activations */
#define SX_ACT_NONE 0
#define SX_ACT_GELU 1 /* exact erf GELU  == torch.nn.GELU()            (qwen_visual.py:253-255) */
#define SX_ACT_SILU 2 /* x*sigmoid(x)   == ACT2FN["silu"]             (modeling_llama_xformer.py:152-167) */

/* A-operand addressing modes of sx_gemm */
#define SX_A_LINEAR 0  /* A is [M][K] row-major                                                   */
#define SX_A_CONV3X3 1 /* A is an NHWC image, implicit im2col of a 3x3 / pad 1 convolution        */

const char* sx_last_error(void);
int sx_version(void);

/* ------------------------------------------------------------------------------------------------
 * GEMM / implicit-GEMM conv on MFMA:  C[M][N] = epilogue( A[M][K] · W[N][K]^T )
 * replaces: nn.Linear everywhere on the path (qwen_visual.py:176-177,253-255,115;
 *           modeling_llama_xformer.py:160-167,184-187,707; resampler.py:12-15,42-44,237,257-258),
 *           nn.Conv2d 3x3 inside diffusers UNet2DConditionModel [ext] (call site
 *           pipeline_stable_diffusion_xl_t2i_edit.py:915-922), F.interpolate(nearest 2x)+conv and the
 *           stride-2 Downsample2D conv of the same UNet.
 * epilogue order: v = acc; v += bias[n]; v += bias2d[(m / bias2d_rows)][n]; if(glu) v = first*act(second)
 *                 else v = act(v); v += residual[(res_mod? m % res_mod : m)][n_out]; store as out_dtype.
 * glu: W rows are packed in 32-row groups [16 "linear" rows | 16 "gate" rows]; output has N/2 columns.
 * ------------------------------------------------------------------------------------------------ */
typedef struct sx_gemm_args {
  const void* A;         /* 16-bit, LINEAR: [M][K]; CONV3X3: [B][Hin][Win][Cin]                    */
  const void* W;         /* 16-bit, [N][K]   (CONV3X3: K = 9*Cin ordered (ky,kx,cin))              */
  void* C;               /* [M][N_out] of out_dtype, row stride ldc elements                        */
  const float* bias;     /* [N] fp32 or NULL                                                        */
  const float* bias2d;   /* [M / bias2d_rows][N] fp32 or NULL  (per-sample time-embedding add)      */
  const float* residual; /* fp32 [.][N_out] row stride ldr, or NULL                                 */
  int32_t M, N, K;
  int32_t ldc, ldr;
  int32_t n_valid;     /* 0 = all; else only output columns < n_valid are stored (multiple of 4)    */
  int32_t res_mod;     /* 0 = residual row m; >0 = residual row m % res_mod (pos-embed broadcast)    */
  int32_t bias2d_rows; /* rows of C per bias2d row                                                   */
  int32_t dtype;       /* SX_F16 / SX_BF16 : type of A and W                                         */
  int32_t out_dtype;   /* SX_F16 / SX_BF16 / SX_F32                                                  */
  int32_t act;         /* SX_ACT_*                                                                   */
  int32_t glu;         /* 0/1                                                                        */
  int32_t a_mode;      /* SX_A_*                                                                     */
  /* conv geometry (a_mode == SX_A_CONV3X3): M = B*Hout*Wout, K = 9*Cin */
  int32_t B, Hin, Win, Cin, Hout, Wout;
  int32_t stride;   /* 1 or 2 (padding: see pad_mode)                                                */
  int32_t upsample; /* 1 = input is nearest-2x upsampled on the fly (Hout = 2*Hin)                   */
  int32_t ld_bias2d; /* row stride of bias2d in floats (0 = N): lets one GEMM produce every resnet's time add    */
  int32_t pad_mode; /* 0 = zero pad 1 on every side; 1 = pad 0 top/left and 1 bottom/right (diffusers Downsample2D with
                       padding=0 + F.pad(x, (0,1,0,1)): the stride-2 convs of the VAE encoder [ext])                 */
  int32_t a_planes; /* 0 / 1: A is [M][K]. 2 (SX_A_LINEAR only): A is [M][2K] = [hi(K) | lo(K)], the two 16-bit planes of an
                       fp32-grade activation x = hi + lo (sx_split16 / sx_rmsnorm_planes / sx_attention_f32 write them); W stays
                       [N][K] and its k-tiles are walked twice: C = epilogue((hi + lo) W^T), fp32 accumulation, no extra weight
                       bytes. The Llama decoder's precise mode (modeling_llama_xformer.py:204-206,239,166-167,707 at 1e-3 of fp32) */
} sx_gemm_args;
int sx_gemm(const sx_gemm_args* args, void* stream);
/* sx_gemm + the statistics pass of the GroupNorm that reads its output (diffusers ResnetBlock2D.norm2 / Transformer2DModel.norm /
 * conv_norm_out [ext] after conv1 / conv2 / proj_out; call site pipeline_stable_diffusion_xl_t2i_edit.py:915-922), fused into the
 * GEMM epilogue: stats[row / rows_per_sample][column / (N / groups)][2] (fp64, host-zeroed or pre-accumulated) += (sum, sum of
 * squares) of the stored fp32 output. Only the 256-row ping-pong tiles carry it: *fused_host = 1 if this launch accumulated,
 * 0 if the GEMM ran unchanged (then run sx_groupnorm's own statistics pass). */
int sx_gemm_gn(const sx_gemm_args* args, double* stats, int groups, int rows_per_sample, int* fused_host, void* stream);
/* A LayerNorm folded into the GEMMs either side of it (reference: every `norm1/2/3 → to_q / to_k / to_v / ff.net[0]` pair of the SDXL
 * transformer blocks [ext BasicTransformerBlock], called per step from pipeline_stable_diffusion_xl_t2i_edit.py:915-922).
 *   PRODUCER launch (row_stats_out != NULL; plain fp32 output, e.g. the attention out-projection + residual): besides C it stores
 *     x16_out[M][ld_x16] = C rounded to the operand dtype and adds every row's (sum, sum of squares) over this launch's columns to
 *     row_stats_out[M][2] (fp64 atomics; zero beforehand).
 *   CONSUMER launch (row_stats_in != NULL; A = that x16 copy, W = the weight with gamma folded in, bias = b + W beta):
 *     out = epilogue(rstd_m * (A W'^T - mu_m * colsum) + bias), mu / rstd of row m from row_stats_in (complete), colsum[n] = sum_k W'[n][k].
 * Ping-pong 256-row tiles only: fails (SX_ERR_INVALID) for a shape the cost model gives a lock-step tile — sx_gemm_pick_tile(...) >= 7
 * says beforehand; such shapes keep the separate sx_layernorm launch. */
typedef struct sx_gemm_ln_args {
  void* x16_out;               /* producer */
  double* row_stats_out;
  const double* row_stats_in;  /* consumer */
  const float* colsum;
  int32_t ld_x16;
  int32_t dim;                 /* LayerNorm width (= K of the consumer) */
  float eps;
  int32_t reserved;
} sx_gemm_ln_args;
int sx_gemm_ln(const sx_gemm_args* args, const sx_gemm_ln_args* ln, void* stream);
/* tuning/test hook: force tile config 0..8 (lock-step 128x128, 128x80, 64x128, 64x64, 256x256, 256x320, 256x160; ping-pong
 * 256x256, 256x320 — the two-wave-group schedule of csrc/gemm_pp.hip); -1 = automatic (cost model); 100/101 = 2-D XCD
 * partition off/on; 200/201 = ping-pong tiles excluded from / offered to the cost model; 300+g = g tile-rows per in-XCD
 * traversal group (300 = default); 400+v = ping-pong schedule variant v (A/B builds of the bf16 256x256 linear kernel);
 * 9 = the persistent strip kernel of csrc/gemm_strip.hip for sx_gemm_ln producers (fails if the launch does not fit it);
 * 500/501 = strip kernel excluded from / offered to the automatic choice; 600 + mask = epilogue A/B switches (bit 0: the GLU
 * epilogue keeps its 8-byte stores) */
int sx_gemm_force_tile(int cfg);
/* tuning hook: `buf` = device buffer of 4 x uint64 per workgroup; following sx_gemm launches store s_memtime stamps
 * {start, first k-tile landed, main loop done, end} per workgroup (tools/gemm_phase_probe.py). NULL switches it off. */
int sx_gemm_debug_stamps(void* buf);
/* host-only query (no launch): tile config 0..8 the cost model picks for an M x N x K problem (glu / conv3x3 flags) */
int sx_gemm_pick_tile(int M, int N, int K, int glu, int conv);

/* 1..32-row GEMV for single-token decode of up to 32 lock-step sequences (HBM-bound weight streaming).
 * replaces: the same nn.Linear calls at q_len == 1 (modeling_llama_xformer.py:204-206,239,166-167,707).
 * y[m][n_out] = epi( x[m][K] · W[N][K]^T ), x 16-bit, y out_dtype, residual fp32. glu packing as above.
 * M <= 4 (or K % 64 != 0 / N % 32 != 0, then M <= 8): one wave per 2 rows, VALU dot products.
 * M >= 5: the weight rows feed a 16x16x32 MFMA against x^T padded to 16 columns (cost independent of M); M = 17..32: two such
 * column blocks per weight fragment — the weights still stream once (tiled operands: [2][K/32][16][32], rows 16..31 in block 1). */
typedef struct sx_gemv_args {
  const void* x;
  const void* W;
  void* y;
  const float* residual;
  int32_t M, N, K;
  int32_t dtype, out_dtype, act, glu;
  int32_t w_layout;    /* 0: W row-major [N][K]. 1: decode tiles [N/16][K/32][16][32] (each 16-row x 32-k MFMA operand tile
                        * is 1 KB contiguous, tiles of a row group follow each other along K): MFMA path only (M >= 2).
                        * 2: 20-row decode tiles [N/20][K/32][20][32] (no GLU, N % 20 == 0): one workgroup per 20 rows — for
                        * N = 5120 that is 256 equal workgroups on the 256 CUs instead of 320 16-row groups (o / down projections) */
  int32_t x_layout;    /* 0: x row-major [M][K]. 1: operand tiles [K/32][16][32] (tile t holds x[0..15][32t .. 32t+31], rows >= M
                        * are padding with ARBITRARY content (uninitialised is fine, NaN bit patterns included): an MFMA output column depends
                        * only on its own operand column and columns >= M are never stored; 16 * K elements in all): MFMA path only */
  void* workspace;     /* optional, MFMA path: device scratch for split-K over workgroups (shapes whose N / 16 row groups do not
                        * fill the chip). Layout: 16 KB of arrival counters, then the partial sums. Must be ZERO when first used
                        * and is left with its counters at zero; bytes >= 16384 + 8 * 16 * N * 4 allows every split factor
                        * (smaller: no split). One launch at a time per
                        * workspace (launches on one stream are fine). NULL: never split. Results are deterministic either way
                        * (partials are added in split order by the last workgroup to arrive). */
  uint64_t workspace_bytes;
  /* RMSNorm folded into the decode step's skinny GEMMs (MFMA path only; all NULL / 0 = off). LlamaRMSNorm (modeling_llama_xformer.py:95,
   * 286, 301) is y = x * rsqrt(mean(x^2) + eps) * gamma: gamma is folded into the NEXT projection's weights at load time and
   * rsqrt(...) is a per-row scalar, so the norm needs no pass of its own —
   *   producer (a GEMV with an fp32 residual output = the new residual stream x): also stores x as 16-bit operand tiles
   *     [N/32][16][32] (`x16_out`, the next GEMV's x with x_layout = 1) and, per workgroup p, the rows' sums of squares over its own
   *     columns (`row_ssq_out`[16][parts], parts = the launch's workgroups in x = sx_gemv_ssq_parts(N, glu, w_layout); the consumer needs parts % 64 == 0: 320 / 256 for N = 5120);
   *   consumer: multiplies its accumulators by rsqrt(sum_p row_ssq_in[m][p] / ssq_dim + ssq_eps) before activation / GLU / store
   *     (the partials are added in the fixed order p = 0, 1, ...: every workgroup computes the same scale). */
  void* x16_out;
  float* row_ssq_out;
  const float* row_ssq_in;
  int32_t ssq_in_parts, ssq_dim;
  float ssq_eps;
  int32_t x_planes;    /* 0 / 1: x is one 16-bit activation. 2 (x_layout 1, MFMA path at any M >= 1): x holds the two planes of an
                        * fp32-grade activation as operand blocks [2 planes][RB][K/32][16][32], RB = ceil(M / 16) row blocks of 16 rows —
                        * the hi plane's row blocks, then the lo plane's (sx_split16 / sx_rmsnorm_planes / sx_attention_f32 with
                        * SX_TILED16): every weight fragment feeds 2 RB MFMAs (M <= 16: two, M = 17..32, round 6: four) and the lo
                        * products are added to the hi products ahead of the epilogue — the weights stream once, y = epi((hi + lo) W^T) */
  int32_t out_planes;  /* 1: the tiled 16-bit output (SX_TILED16) or x16_out is written as two planes [2][RB][cols/32][16][32]
                        * (hi = rn16(v), lo = rn16(v - hi)): the next sx_gemv's x with x_planes = 2, without a sx_split16 launch */
  const float* x16_gamma; /* optional fp32 [N] with x16_out: x16_out holds o * gamma (the NEXT LlamaRMSNorm's weight applied on the
                        * activation side, so that the next projection keeps its exact checkpoint weights and only scales by rstd from
                        * row_ssq_in — the RMSNorm fold of the precise mode; row_ssq_out is still the sum of squares of o itself) */
} sx_gemv_args;
/* workgroups in x (= partial rows of row_ssq_out) sx_gemv launches for an M x N x K problem with / without GLU on the MFMA path */
int sx_gemv_ssq_parts(int N, int glu, int w_layout);
int sx_gemv(const sx_gemv_args* args, void* stream);
/* test hook: 1 = always take the VALU path (lets the tests compare both), 0 = automatic */
int sx_gemv_force_valu(int on);
/* tuning hook (tools/lab/gemv_lab): key 2 = split-K factor of the MFMA skinny GEMM when a workspace is given:
 * 0 automatic, -1 / 1 never, 2 / 4 / 8 forced; key 1 = 1: the 17..32-row kernels with 4 k-steps per round (default 2) */
int sx_gemv_tune(int key, int value);

/* ------------------------------------------------------------------------------------------------
 * Normalisations (row reductions with wave shuffles, fp32 statistics)
 * ------------------------------------------------------------------------------------------------ */
/* LayerNorm over the last dim. replaces nn.LayerNorm (qwen_visual.py:246,250,361,384,122-123;
 * resampler.py:11,38-39,240) and diffusers BasicTransformerBlock.norm1/2/3 [ext].
 * rms != 0 → LlamaRMSNorm (modeling_llama_xformer.py:95,286,301,595): y = x * rsqrt(mean(x^2)+eps) * gamma */
int sx_layernorm(const void* x, int in_dtype, void* y, int out_dtype, const float* gamma, const float* beta,
                 int rows, int cols, float eps, int rms, void* stream);

/* Row softmax y = softmax(scale * x) over the last dim, fp32 in → 16-bit out (probabilities feed the P·V GEMM) or fp32
 * out (SX_F32, ldy in floats: the fp32-grade VAE mode splits it into bf16 planes).
 * replaces: the softmax inside diffusers Attention [ext] of the VAE decoder's mid block (one 512-wide head over
 * (H/8)·(W/8) pixels — head_dim 512 does not fit the flash kernel, so scores go through sx_gemm; reference call site
 * pipeline_stable_diffusion_xl_t2i_edit.py:973). */
int sx_softmax_rows(const float* x, int64_t ldx, void* y, int64_t ldy, int rows, int cols, float scale, int out_dtype,
                    void* stream);

/* GroupNorm(+SiLU) over NHWC activations x[B][HW][C] (fp32 in). replaces diffusers
 * ResnetBlock2D.norm1/norm2 + nonlinearity, Transformer2DModel.norm, conv_norm_out [ext].
 * stats: scratch fp64 [B][groups][2] (zeroed by the call). y: 16-bit normalised output; SX_F32 is accepted when raw16 is
 * NULL; SX_BF16X3 writes y (and raw16) as [B][HW][3*C] bf16 planes, SX_F16X2 as [B][HW][2*C] fp16 planes: the A operands of the
 * fp32-grade VAE mode.
 * raw16: optional 16-bit un-normalised copy of x (feeds the 1x1 shortcut conv), may be NULL. */
int sx_groupnorm(const float* x, void* y, void* raw16, int out_dtype, const float* gamma, const float* beta,
                 double* stats, int B, int HW, int C, int groups, float eps, int silu, void* stream);
/* same with the input given as the channel concatenation [x | x2] of two NHWC tensors (x: [B][HW][C1], x2: [B][HW][C-C1]):
 * GroupNorm over torch.cat([hidden_states, res_hidden_states], dim=1) of the UNet up blocks (diffusers
 * CrossAttnUpBlock2D / UpBlock2D [ext]) without materialising the concatenation. x2 == NULL: plain sx_groupnorm. */
int sx_groupnorm2(const float* x, const float* x2, int C1, void* y, void* raw16, int out_dtype, const float* gamma,
                  const float* beta, double* stats, int B, int HW, int C, int groups, float eps, int silu, void* stream);
/* tuning hook: key 0 = block count of the GroupNorm statistics pass (0 = chosen by input size, the default) */
int sx_norm_tune(int key, int value);
/* the same split in two calls for pixel-sharded (sequence-parallel) UNet ranks: phase 1 = zero `stats` and accumulate the
 * fp64 sum / sum-of-squares of THIS rank's HW rows per (sample, group); the caller all-reduces `stats` across ranks;
 * phase 2 = normalise this rank's rows with statistics that cover hw_total rows per sample. */
int sx_groupnorm_sp(const float* x, const float* x2, int C1, void* y, void* raw16, int out_dtype, const float* gamma,
                    const float* beta, double* stats, int B, int HW, int hw_total, int C, int groups, float eps, int silu,
                    int phase, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Attention
 * ------------------------------------------------------------------------------------------------ */
/* Flash-style fused attention on MFMA (online softmax in fp32, LDS-staged K / V^T tiles).
 * replaces: bmm→softmax→bmm (qwen_visual.py:204-215), nn.MultiheadAttention core (qwen_visual.py:145),
 *           xformers memory_efficient_attention (modeling_llama_xformer.py:221-238),
 *           torch SDPA in diffusers AttnProcessor2_0 [ext].
 * Q[b][sq][h][D], K[b][skv][h][D], V[b][skv][h][D] addressed with element strides (multiples of 8 elements).
 * O[b][sq][h][D] (o_row_stride = elements between consecutive sq; heads packed at h*D).
 * causal: key j visible to query i iff j <= i + (Skv - Sq)  (bottom-right aligned; prefill and chunked decode).
 * D must be a multiple of 8 and <= 128 (104 is padded to 128 in LDS/registers only). */
typedef struct sx_attn_args {
  const void* Q;
  const void* K;
  const void* V;
  void* O;
  int32_t B, H, Sq, Skv, D, reserved;
  int64_t q_batch_stride, q_row_stride, q_head_stride;
  int64_t k_batch_stride, k_row_stride, k_head_stride;
  int64_t v_batch_stride, v_row_stride, v_head_stride;
  int64_t o_batch_stride, o_row_stride;
  float scale;
  int32_t causal;
  int32_t dtype;
} sx_attn_args;
int sx_attention(const sx_attn_args* args, void* stream);
/* tuning hook (tools/lab/attn_lab): A/B builds of the flash kernel; 0 = shipped */
int sx_attention_variant(int v);

/* Small generic attention (any D <= 256, VALU, one wave per query row). Used where FLOPs are negligible:
 * Resampler MHA with head_dim 160 (agent_seed_x_i.yaml:2-7), AttentionPool2d (resampler.py:89-116),
 * PerceiverAttention (resampler.py:46-75). V is in natural [b][skv][h][D] layout. */
typedef struct sx_attn_small_args {
  const void* Q;
  const void* K;
  const void* V;
  void* O;
  int32_t B, H, Sq, Skv, D;
  int64_t q_batch_stride, q_row_stride, q_head_stride;
  int64_t k_batch_stride, k_row_stride, k_head_stride;
  int64_t v_batch_stride, v_row_stride, v_head_stride;
  int64_t o_batch_stride, o_row_stride;
  float scale;
  int32_t dtype;
} sx_attn_small_args;
int sx_attention_small(const sx_attn_small_args* args, void* stream);

/* Single-token decode attention over the KV cache (split-KV flash-decoding, HBM-bound).
 * replaces memory_efficient_attention at q_len == 1 (modeling_llama_xformer.py:231-237).
 * q[H][D] 16-bit; kcache/vcache [H][Tmax][D]; ctx_len read from DEVICE memory (graph-replay friendly);
 * scratch: fp32 [H][nsplit][D+2]; out[H*D] 16-bit. */
int sx_attn_decode(const void* q, const void* kcache, const void* vcache, void* out, float* scratch,
                   const int32_t* ctx_len_dev, int H, int D, int Tmax, int nsplit, float scale, int dtype,
                   void* stream);

/* ------------------------------------------------------------------------------------------------
 * LLM glue kernels
 * ------------------------------------------------------------------------------------------------ */
/* RoPE (HF rotate-half, tables rounded to the activation dtype first — modeling_llama_xformer.py:128-149)
 * applied in place to the q part of qkv[T][3*H*D] and written, with v, into the caches at pos0+t.
 * pos0 read from device memory. */
int sx_rope_kv_append(void* qkv, void* kcache, void* vcache, const float* cos_tab, const float* sin_tab,
                      const int32_t* pos0_dev, int T, int H, int D, int Tmax, int dtype, void* stream);
/* out[t][:] = table[ids[t]][:]   (embed_tokens, seed_x.py:158) ; out fp32 */
int sx_embedding(const int32_t* ids, const void* table, float* out, int T, int dim, int dtype, void* stream);
/* dst[rows[i]][:] = src[i][:] fp32 (input_embeds[ids_cmp_mask] = ..., seed_x.py:173) */
int sx_scatter_rows(const float* src, const int32_t* rows, float* dst, int n, int dim, void* stream);
/* Greedy step with the AutoImageTokenGenerationProcessor rule fused (generation.py:19-31):
 * prev = *prev_id_dev; if prev in img_ids[0..n_img-2] → next = img_ids[idx+1] (the "max+10" rule) else
 * logits[img_ids[1..]] = 0.0 (in place, as the reference does) then argmax (first maximal index).
 * Writes next to *next_id_dev (may alias prev_id_dev) and, if out_ids != NULL, to out_ids[*step_dev]. */
int sx_greedy_next(float* logits, int vocab, const int32_t* img_ids_dev, int n_img, const int32_t* prev_id_dev,
                   int32_t* next_id_dev, int32_t* out_ids, const int32_t* step_dev, void* stream);

/* Lock-step batched decode (G independent sequences, one token each): same kernels with a sequence dimension.
 * Caches are [G][H][Tmax][D] (cache_seq_stride elements apart), positions / context lengths / step counters / token ids
 * are per-sequence device arrays, so weights are streamed from HBM once per step for all G sequences. */
int sx_rope_kv_append_b(void* qkv, void* kcache, void* vcache, const float* cos_tab, const float* sin_tab,
                        const int32_t* pos0_dev, int G, int T, int H, int D, int Tmax, int64_t cache_seq_stride, int dtype,
                        void* stream);
/* q_seq_stride: elements between the q rows of consecutive sequences (0 = H * D, contiguous) — 3 * H * D reads q straight out of
 * the fused [G][3 * H * D] qkv projection. dtype may carry SX_TILED16 (the output as operand tiles for the o-projection). */
int sx_attn_decode_b(const void* q, const void* kcache, const void* vcache, void* out, float* scratch,
                     const int32_t* ctx_len_dev, int G, int H, int D, int Tmax, int64_t cache_seq_stride, int nsplit,
                     float scale, int dtype, int64_t q_seq_stride, void* stream);
/* One launch per layer for the lock-step decode step: RoPE of the new token's q and k, K / V append at pos[g], split-KV attention
 * over the pos[g] + 1 visible keys and the combine of the splits. Replaces sx_rope_kv_append_b (T = 1) + sx_attn_decode_b with
 * bit-identical output and cache contents (modeling_llama_xformer.py:204-239 at q_len == 1). q is NOT rotated in place. */
typedef struct sx_attn_decode_args {
  const void* qkv;          /* [G][3*H*D] 16-bit rows q | k | v of the new token (output of the fused qkv projection)      */
  void* kcache;             /* [G][H][Tmax][D], sequences cache_seq_stride elements apart                                   */
  void* vcache;
  void* out;                /* [G][H*D] 16-bit, or operand tiles with dtype | SX_TILED16                                    */
  float* scratch;           /* [G][H][nsplit][D + 2] fp32                                                                   */
  void* counters;           /* uint32 [G*H]: zero when first used, left at zero                                             */
  const float* cos_tab;     /* [Tmax][D/2] fp32                                                                             */
  const float* sin_tab;
  const int32_t* pos_dev;   /* [G] position of the new token; positions outside [0, Tmax) attend to the cache, write nothing */
  int32_t G, H, D, Tmax, nsplit, dtype;
  int64_t cache_seq_stride;
  float scale;
  int32_t reserved;
} sx_attn_decode_args;
int sx_attn_decode_fused(const sx_attn_decode_args* args, void* stream);
int sx_greedy_next_b(float* logits, int ld_logits, int vocab, const int32_t* img_ids_dev, int n_img,
                     const int32_t* prev_id_dev, int32_t* next_id_dev, int32_t* out_ids, int ld_out,
                     const int32_t* step_dev, int G, void* stream);
/* dst[(g*seq_rows + step[g])][:] = src[g][:] fp32 (per-sequence hidden-state log, seed_x.py:196) */
int sx_scatter_rows_step(const float* src, const int32_t* step_dev, float* dst, int G, int dim, int seq_rows, void* stream);
/* p[0..n) += delta */
int sx_add_i32_n(int32_t* p, int delta, int n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Elementwise / layout helpers
 * ------------------------------------------------------------------------------------------------ */
int sx_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream);
/* fp32 x[rows][cols] → bf16 out[rows][3*cols] holding the planes of x = hi + lo (hi = bf16(x), lo = bf16(x - hi)):
 * role 0 (A operand rows) = [hi | hi | lo], role 1 (W operand rows) = [hi | lo | hi], so that one sx_gemm with K = 3*cols
 * computes Ah·Wh + Ah·Wl + Al·Wh with fp32 accumulation. cols % 4 == 0. Operand preparation of the VAE's fp32-grade mode:
 * stands in for the fp32 VAE the reference's pipeline switches to in upcast_vae()
 * (pipeline_stable_diffusion_xl_t2i_edit.py:509-511, :569-586, :965-977). */
int sx_split_bf16(const float* x, void* out, int64_t rows, int cols, int role, void* stream);

/* ------------------------------------------------------------------------------------------------
 * fp32-grade activations of the Llama decoder (LlamaForCausalLM(precise=True); csrc/precise.hip): a 16-bit checkpoint's weights
 * are exact, so logits within 1e-3 of the fp32 reference at 40 layers only need more mantissa on the ACTIVATION side:
 * GEMM A operands travel as two 16-bit planes x = hi + lo (hi = rn16(x), lo = rn16(x - hi)), q / k / v and the KV cache stay fp32.
 * `dtype` below = SX_F16 / SX_BF16 of the planes; | SX_TILED16 → operand tiles [2 planes][ceil(rows / 16)][cols/32][16][32] (the hi
 * plane's row blocks, then the lo plane's; rows <= 32: the x operand of sx_gemv with x_planes = 2) instead of rows [rows][2*cols] =
 * [hi | lo] (sx_gemm with a_planes = 2).
 * ------------------------------------------------------------------------------------------------ */
/* fp32 x[rows][cols] (row stride ldx) → the two planes. cols % 8 == 0 (tiles: cols % 32 == 0). */
int sx_split16(const float* x, int64_t ldx, void* out, int rows, int cols, int dtype, void* stream);
/* LlamaRMSNorm (modeling_llama_xformer.py:95,286,301,595): y = gamma * (x * rsqrt(mean(x^2) + eps)), all fp32; y32 (fp32 [rows][cols],
 * may be NULL) and / or the planes of y (out16, may be NULL). */
int sx_rmsnorm_planes(const float* x, const float* gamma, float* y32, void* out16, int rows, int cols, float eps, int dtype,
                      void* stream);
/* sx_rope_kv_append_b on fp32 rows and fp32 caches (modeling_llama_xformer.py:141-149,215-220): qkv fp32 [G*T][3*H*D], q rotated in
 * place, rotated k and v appended to kcache / vcache fp32 [G][H][Tmax][D] at pos0[g] + t. The cos / sin tables are rounded to
 * table_dtype first (:128-131 casts them to the activation dtype); positions outside [0, Tmax) write nothing. */
int sx_rope_kv_append_f32(float* qkv, float* kcache, float* vcache, const float* cos_tab, const float* sin_tab,
                          const int32_t* pos0_dev, int G, int T, int H, int D, int Tmax, int64_t cache_seq_stride,
                          int table_dtype, void* stream);
/* the same with the "mixed" cache: k fp32, v appended in table_dtype (16-bit [G][H][Tmax][D], the same element strides): three
 * quarters of the cache bytes; v's rounding only perturbs the softmax-weighted average */
int sx_rope_kv_append_f32_v16(float* qkv, float* kcache, void* vcache16, const float* cos_tab, const float* sin_tab,
                              const int32_t* pos0_dev, int G, int T, int H, int D, int Tmax, int64_t cache_seq_stride,
                              int table_dtype, void* stream);
/* Causal attention of a T-token chunk per sequence over the fp32 cache, fp32 FMA arithmetic and softmax
 * (modeling_llama_xformer.py:204-239: prefill causal, q_len == 1 sees the whole cache): row t sees keys 0 .. pos0[g] + t.
 * Device-resident positions → graph-capturable for the decode step (T = 1). Output = the planes of the context rows [G*T][H*D].
 * causal = 0: every row sees all Tmax keys (pos0_dev unused) — with the K / V strides below this is also the small fp32 attention of
 * ResamplerXLV2's precise mode (PerceiverAttention / AttentionPool2d, resampler.py:42-87,89-116: fp32 softmax over <= 129 keys). */
typedef struct sx_attn_f32_args {
  const float* q;           /* rotated q: row g*T + t at q + row*q_row_stride, head h at + h*D (e.g. the qkv buffer, stride 3*H*D) */
  const float* kcache;      /* fp32 [G][H][Tmax][D], sequences cache_seq_stride floats apart                                       */
  const void* vcache;       /* fp32, or — v16 = 1 — the "mixed" cache's 16-bit V (the planes' dtype) with the same ELEMENT strides     */
  void* out;                /* planes of [G*T][H*D], layout by dtype (see above)                                                    */
  const int32_t* pos0_dev;  /* [G] cache position of each sequence's first chunk token (causal)                                    */
  int64_t q_row_stride, cache_seq_stride;
  int64_t kv_row_stride;    /* floats between consecutive keys of a head (0 = D)                                                   */
  int64_t kv_head_stride;   /* floats between heads (0 = Tmax*D: the cache layout)                                                  */
  int32_t G, T, H, D, Tmax, dtype;
  float scale;
  int32_t causal;           /* 1: row t sees keys 0 .. pos0[g] + t; 0: all Tmax keys                                               */
  int32_t v16;              /* 0: fp32 V (default); 1: V is 16-bit, in the planes' dtype (head_dim <= 128): the mixed cache          */
  int32_t nsplit;           /* T == 1 only: > 1 spreads the keys of every (head, sequence) over nsplit workgroups + a combine launch   */
  float* scratch;           /* nsplit > 1: fp32 [G][H][nsplit][D + 2] partial results                                                */
  /* T == 1, causal, D == 128 only — RoPE + KV append fused into the step (modeling_llama_xformer.py:141-149,204-244 in ONE launch
   * per layer): rope_cos != NULL means q is the UNROTATED row, k_new / v_new are the new token's rows (row stride q_row_stride: the
   * qkv buffer's + H*D and + 2*H*D), and the launch appends the rotated k and v (rounded when v16) to kcache / vcache at pos0[g] —
   * the two cache pointers are written through in this form. NULL (default) = sx_rope_kv_append_f32* ran before. */
  const float* rope_cos;    /* [Tmax][D/2] fp32 (rounded to the planes' dtype inside, like sx_rope_kv_append_f32)                    */
  const float* rope_sin;
  const float* k_new;
  const float* v_new;
} sx_attn_f32_args;
int sx_attention_f32(const sx_attn_f32_args* args, void* stream);
/* tuning / test hook: 1 (default) = causal chunks above 8 tokens at head_dim 128 run on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32)
 * flash kernel; 0 = the VALU kernels everywhere (the bit-reference of its test, and the A/B) */
int sx_attention_f32_variant(int v);
/* strided 2-D copy of fp32 rows: dst[r][dst_off + c] = src[r][c]  (channel concat of skip connections) */
int sx_copy2d_f32(const float* src, int64_t src_ld, float* dst, int64_t dst_ld, int64_t rows, int cols,
                  void* stream);
/* y = a + b (fp32), n elements */
int sx_add_f32(const float* a, const float* b, float* y, int64_t n, void* stream);
/* ViT patchify: img[B][3][S][S] (fp32) → patches[B*(S/P)^2][Kpad] 16-bit, k = c*P*P + py*P + px, zero padded
 * (conv1 as a GEMM, qwen_visual.py:352,393-396) */
int sx_patchify(const float* img, void* patches, int B, int S, int P, int Kpad, int dtype, void* stream);
/* generic 3x3/pad1 im2col for tiny Cin (UNet conv_in, Cin = 4|8): x[B][H][W][Cin] fp32 → [B*H*W][Kpad] */
int sx_im2col3x3_small(const float* x, void* out, int B, int H, int W, int Cin, int Kpad, int dtype,
                       void* stream);
/* avg_pool1d(k=4,s=4) over tokens: x[B][L][D] → y[B][L/4][D] fp32 (adapter_modules.py:112-115) */
int sx_avgpool_tokens(const float* x, float* y, int B, int L, int D, int k, void* stream);
/* sinusoidal timestep embedding, flip_sin_to_cos=True, shift 0 (diffusers Timesteps [ext]):
 * out[i][:] = [cos(t_i*f_j) | sin(t_i*f_j)], f_j = exp(-ln(10000) * j / half); t_i = t[i], or t[*idx_dev] for every
 * row when idx_dev != NULL (denoise-step counter kept on the device so the step is graph-replayable) */
int sx_timestep_embedding(const float* t, const int32_t* idx_dev, void* out, int n, int dim, int dtype, void* stream);
/* NCHW fp32 ↔ NHWC fp32 for the 4/8-channel latents; ld = channel stride of the NHWC side */
int sx_nchw_to_nhwc(const float* src, float* dst, int ld, int B, int C, int HW, void* stream);
int sx_nhwc_to_nchw(const float* src, int ld, float* dst, int B, int C, int HW, void* stream);
/* *p += delta on the device (step / position counters of graph-replayed loops) */
int sx_add_i32(int32_t* p, int delta, void* stream);
/* profiling hook: one empty dispatch of `profile_marker_kernel` (tools/kstats_step.py cuts a rocprofv3 kernel trace at these) */
int sx_profile_marker(int tag, void* stream);
/* y = silu(x) as 16-bit (ResnetBlock2D.time_emb_proj input: nonlinearity(temb), diffusers [ext]) */
int sx_silu_cast(const float* x, void* y, int dtype, int64_t n, void* stream);

/* Fused classifier-free guidance + Euler step on fp32 latents (NHWC [1][HW][C], n = HW*C; eps [nb][HW][C]).
 * mode 0 (t2i, StableDiffusionXLPipeline.__call__ [ext]; order [uncond, text]):
 *     eps = e0 + gs*(e1 - e0);  lat += eps * (sigma_next - sigma)
 * mode 1 (edit, pipeline_stable_diffusion_xl_t2i_edit.py:926-953; order [text, image, uncond]):
 *     x0_k = lat - sigma*e_k;  x0 = x0_u + gs*(x0_t - x0_i) + igs*(x0_i - x0_u);
 *     eps = (x0 - lat)/(-sigma);  lat += eps * (sigma_next - sigma)
 * Also writes the next step's scaled model input: scaled[k][hw][c] = lat_new / sqrt(sigma_next^2 + 1), k < nb,
 * into an NHWC buffer with channel stride ld_scaled (8 for the edit UNet: channels 4..7 hold the image latents).
 * sigmas live on the device (sigmas_dev[step], sigmas_dev[step+1]); step read from *step_dev. */
int sx_cfg_euler_step(const float* eps, float* latents, float* scaled_next, const float* sigmas_dev,
                      const int32_t* step_dev, int nb, int64_t n, int C, int ld_scaled, float gs, float igs, int mode,
                      void* stream);

/* ------------------------------------------------------------------------------------------------
 * Image pre/post-processing: the byte work either side of the dense paths.
 * ------------------------------------------------------------------------------------------------ */
/* Pillow `Image.resize` for 8-bit images (ImagingResample, Resample.c [ext Pillow]), bit exact: separable antialiased
 * convolution in 22-bit fixed point, horizontal pass (→ uint8) then vertical pass (→ uint8).
 * replaces: `image.resize(...)` (default BICUBIC) at src/inference/any_res.py:104,111,183,187 and torchvision
 * `transforms.Resize` on PIL input (BILINEAR) at src/processer/transforms.py:8-14.
 * src: [Hin][Win][C] uint8, row stride src_stride bytes; dst: dense [Hout][Wout][C] uint8. C = 1 | 3.
 * kk_h / bounds_h (kk_v / bounds_v): Pillow's normalize_coeffs_8bpc(precompute_coeffs(...)) tables built by the host:
 * kk[out][ksize] int32, bounds[out][2] = {first input index, count}. A NULL kk_* skips that pass (Pillow does the same
 * when the size along that axis is unchanged). With both passes, `tmp` holds the horizontally resampled rows
 * [y_first, y_first + y_rows) — the rows the vertical pass touches — as dense [y_rows][Wout][C] uint8. */
int sx_resample_u8(const void* src, int Hin, int Win, int C, int64_t src_stride, void* dst, int Hout, int Wout,
                   const int32_t* kk_h, const int32_t* bounds_h, int ksize_h, const int32_t* kk_v,
                   const int32_t* bounds_v, int ksize_v, int y_first, int y_rows, void* tmp, void* stream);
/* crop + ToTensor + Normalize in one gather: dst[c][y][x] = lut3x256[c][src[y0+y][x0+x][c]], dst fp32 [3][Hc][Wc].
 * replaces: `image.crop(box)` (any_res.py:130-134), transforms.ToTensor() and transforms.Normalize(mean, std)
 * (src/processer/transforms.py:16-19). The host builds lut[c][v] = (float32(v)/255 - mean[c]) / std[c] in float32. */
int sx_u8_to_chw_lut(const void* src, int H, int W, int64_t src_stride, int x0, int y0, int Hc, int Wc,
                     const float* lut3x256, float* dst, void* stream);
/* VaeImageProcessor.postprocess(output_type="pil") [ext diffusers] as called at
 * pipeline_stable_diffusion_xl_t2i_edit.py:986: src fp32 [3][H][W] → dst uint8 [H][W][3],
 * u = round_half_even(clamp(x/2 + 0.5, 0, 1) * 255). */
int sx_chw_to_u8_image(const float* src, int H, int W, void* dst, void* stream);
/* ids_cmp_mask of eval_img2text_seed_x_i.py:153-160: mask[i] = 1 strictly between the k-th opening marker (boi | bop)
 * and the k-th closing marker (eoi | eop), pairs formed like zip(boi_indices, eoi_indices). ids: int64 [T]; mask: uint8 [T]. */
int sx_marker_mask(const int64_t* ids, int T, int64_t boi, int64_t bop, int64_t eoi, int64_t eop, void* mask,
                   void* stream);
/* F.normalize(x) with its default dim=1 on x[B][T][D] fp32 (the token axis — ResamplerXLV2(normalize=True),
 * src/models/detokenizer/resampler.py:271-272): y = x / max(||x[b,:,d]||_2, eps). */
int sx_l2norm_dim1(const float* x, float* y, int B, int T, int D, float eps, void* stream);

/* ------------------------------------------------------------------------------------------------
 * One-shot all-reduce / all-gather over peer-mapped device memory (csrc/comm.hip): the tensor-parallel decode step's
 * 20 KB - 320 KB collectives as ONE graph-capturable kernel launch per rank. No reference counterpart (the reference is
 * single-device inference; src/train/dist_utils.py:5-34 is its only collective code): north_star's TP requirement.
 * The payload is cut into chunks of `chunk` floats (SX_ONESHOT_CHUNK unless overridden); one workgroup per chunk runs the
 * whole publish / signal / wait / reduce protocol on its own epoch counter and flag row, all chunks in flight at once.
 * Set-up (once): every rank sx_comm_alloc()s a staging area (2 x cap floats) and a flag array ((cap / chunk) x world x uint32),
 * sx_ipc_export()s both, sends the 64-byte handles to its peers (any transport), sx_ipc_open()s the peers' handles and
 * stores the world pointers in two DEVICE arrays (own buffers at index rank).
 * ------------------------------------------------------------------------------------------------ */
int sx_comm_alloc(void** ptr, uint64_t bytes);          /* zero-initialised device memory that can be exported        */
int sx_comm_free(void* ptr);
int sx_ipc_export(void* ptr, unsigned char* handle64);  /* hipIpcGetMemHandle: 64 opaque bytes                        */
int sx_ipc_open(const unsigned char* handle64, void** ptr);
int sx_ipc_close(void* ptr);
#define SX_ONESHOT_CHUNK 4096 /* floats per workgroup: 16 KB                                                          */
typedef struct sx_oneshot_args {
  void* data;         /* fp32 [n]: all-reduce in place (gather_out == NULL) or the all-gather input                    */
  void* gather_out;   /* fp32 [world][n] or NULL                                                                       */
  const void* stage;  /* DEVICE array of `world` pointers: rank r's staging area                                       */
  const void* flags;  /* DEVICE array of `world` pointers: rank r's flag array                                         */
  void* epoch;        /* this rank's uint32 epoch counters [cap / chunk] (device, zero at start; advanced by the kernel) */
  void* status;       /* uint32 (device): non-zero after a peer failed to arrive within max_spin polls                */
  int32_t n, cap, rank, world;
  uint32_t max_spin;  /* 0 = default (2^22 polls, a few seconds)                                                       */
  int32_t chunk;      /* floats per workgroup, even, divides cap; 0 = SX_ONESHOT_CHUNK. Fixed for the life of the buffers */
  int32_t mode;       /* 0: all-reduce / all-gather among all ranks. 1: NEIGHBOUR exchange (the row-sharded UNet's conv halo rows,
                       * seqpar.py): data = [my last row | my first row] (n floats, n even), gather_out[n] receives [last row of rank - 1 |
                       * first row of rank + 1] (zeros at the ends of the chain); a rank signals and waits for its one or two
                       * neighbours only. Use a communicator of its own for this mode (its epochs must not interleave with mode 0:
                       * a slot is reused once the NEIGHBOURS have moved on) */
  int32_t reserved;
} sx_oneshot_args;
int sx_allreduce_oneshot(const sx_oneshot_args* args, void* stream);

/* Row-sharded 3x3 convolution input (seqpar.with_halo): out[B][Hl + 1 (+1 if bottom)][W + left_col][C] 16-bit = the local slab
 * x[B][Hl][W][C] with the neighbour rows above (prev_row[B][W][C], NULL = zeros: image border) and below (next_row) and, with
 * left_col, a zero column on the left — one launch instead of torch.zeros + three slice copies. C % 8 == 0. */
int sx_halo_pack(const void* x, const void* prev_row, const void* next_row, void* out, int B, int Hl, int W, int C, int left_col,
                 int bottom, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SEEDX_HIP_H */
