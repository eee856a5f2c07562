"""Kernel-level parity on the GPU: every HIP kernel vs a plain PyTorch fp32 reference of the same op computed
from the SAME 16-bit-rounded inputs (so only accumulation order / output rounding differ).

Tolerances (relative L2 error ‖x−ref‖/‖ref‖): fp32 outputs 2e-5; fp16 outputs 6e-4; bf16 outputs 4e-3
(= output rounding, eps_fp16/2 = 4.9e-4, eps_bf16/2 = 3.9e-3 per element)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 2e-5, torch.float16: 6e-4, torch.bfloat16: 4e-3}


def relerr(x, ref):
    x, ref = x.float(), ref.float()
    return ((x - ref).norm() / ref.norm().clamp_min(1e-20)).item()


def rnd(shape, dtype, dev, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).to(dev)


def glu_pack(w_lin, w_gate):
    """[N/2,K] x2 -> [N,K] in 32-row groups [16 linear | 16 gate] (layout contract of sx_gemm glu)."""
    n2, k = w_lin.shape
    a = w_lin.view(n2 // 16, 16, k)
    b = w_gate.view(n2 // 16, 16, k)
    return torch.cat([a, b], dim=1).reshape(2 * n2, k).contiguous()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 128), (200, 144, 192), (77, 80, 64), (1024, 1664, 640),
                                   (2048, 1280, 1280), (64, 512, 256), (333, 4096, 128)])
def test_gemm_plain(dev, dtype, M, N, K):
    from seedx_amd import ops
    a, w = rnd((M, K), dtype, dev, seed=1), rnd((N, K), dtype, dev, 0.05, seed=2)
    ref = a.float() @ w.float().t()
    out32 = ops.gemm(a, w, out_dtype=torch.float32)
    assert relerr(out32, ref) < TOL[torch.float32]
    out16 = ops.gemm(a, w)
    assert out16.dtype == dtype and relerr(out16, ref) < TOL[dtype]


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("M,N,K", [(300, 272, 192), (2048, 1280, 640), (128, 80, 64), (77, 336, 1024)])
def test_gemm_every_tile_config(dev, tile, M, N, K):
    """Each tile shape / pipeline depth (128x128x2, 128x80x3, 64x128x3, 64x64x3) forced in turn, linear and conv."""
    from seedx_amd import _lib, ops
    lib = _lib.load()
    dtype = torch.bfloat16
    a, w = rnd((M, K), dtype, dev, seed=60), rnd((N, K), dtype, dev, 0.05, seed=61)
    bias, res = rnd((N,), torch.float32, dev, seed=62), rnd((M, N), torch.float32, dev, seed=63)
    x = rnd((2, 12, 10, 64), dtype, dev, seed=64)
    wc = rnd((N, 9 * 64), dtype, dev, 0.05, seed=65)
    try:
        lib.sx_gemm_force_tile(tile)
        out = ops.gemm(a, w, bias=bias, residual=res, act="silu", out_dtype=torch.float32)
        outc = ops.conv3x3(x, wc, bias=bias, out_dtype=torch.float32)
    finally:
        lib.sx_gemm_force_tile(-1)
    assert relerr(out, F.silu(a.float() @ w.float().t() + bias) + res) < 5e-5
    assert relerr(outc, _conv_ref(x, wc, bias, 1, False)) < TOL[torch.float32]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_gemm_asymmetric_identity(dev, dtype):
    """A = I with an asymmetric W catches transposed / permuted fragment layouts."""
    from seedx_amd import ops
    K = 128
    a = torch.eye(K, dtype=dtype, device=dev)
    w = (torch.arange(192 * K, device=dev).view(192, K) % 251).to(dtype) / 64
    out = ops.gemm(a, w, out_dtype=torch.float32)
    assert torch.equal(out, w.float().t().contiguous())


@pytest.mark.parametrize("act", [None, "gelu", "silu"])
def test_gemm_epilogues(dev, act):
    from seedx_amd import ops
    dtype = torch.bfloat16
    M, N, K = 300, 272, 128
    a, w = rnd((M, K), dtype, dev, seed=3), rnd((N, K), dtype, dev, 0.1, seed=4)
    bias = rnd((N,), torch.float32, dev, seed=5)
    res = rnd((M, N), torch.float32, dev, seed=6)
    ref = a.float() @ w.float().t() + bias
    if act == "gelu":
        ref = F.gelu(ref)
    elif act == "silu":
        ref = F.silu(ref)
    out = ops.gemm(a, w, bias=bias, act=act, residual=res, out_dtype=torch.float32)
    assert relerr(out, ref + res) < 5e-5
    # residual broadcast (pos-embed): rows m % 100
    res2 = rnd((100, N), torch.float32, dev, seed=7)
    out = ops.gemm(a, w, bias=bias, act=act, residual=res2, res_mod=100, out_dtype=torch.float32)
    assert relerr(out, ref + res2.repeat(3, 1)) < 5e-5
    # per-sample bias2d (time embedding): 3 samples x 100 rows
    b2 = rnd((3, N), torch.float32, dev, seed=8)
    ref2 = a.float() @ w.float().t() + bias + b2.repeat_interleave(100, 0)
    out = ops.gemm(a, w, bias=bias, bias2d=b2, bias2d_rows=100, out_dtype=torch.float32)
    assert relerr(out, ref2) < 5e-5


@pytest.mark.parametrize("act", ["gelu", "silu"])
@pytest.mark.parametrize("M", [200, 4])
def test_gemm_glu(dev, act, M):
    from seedx_amd import ops
    dtype = torch.float16
    N2, K = 160, 192
    a = rnd((M, K), dtype, dev, seed=9)
    wl, wg = rnd((N2, K), dtype, dev, 0.1, seed=10), rnd((N2, K), dtype, dev, 0.1, seed=11)
    bl, bg = rnd((N2,), torch.float32, dev, seed=12), rnd((N2,), torch.float32, dev, seed=13)
    w = glu_pack(wl, wg)
    bias = glu_pack(bl.view(-1, 1), bg.view(-1, 1)).view(-1).contiguous()
    lin = a.float() @ wl.float().t() + bl
    gate = a.float() @ wg.float().t() + bg
    ref = lin * (F.gelu(gate) if act == "gelu" else F.silu(gate))
    out = ops.gemm(a, w, bias=bias, act=act, glu=True, out_dtype=torch.float32)
    assert out.shape == (M, N2) and relerr(out, ref) < 5e-5
    if M <= 8:
        ref_nb = (a.float() @ wl.float().t()) * (F.gelu(a.float() @ wg.float().t()) if act == "gelu"
                                                 else F.silu(a.float() @ wg.float().t()))
        out = ops.gemv(a, w, act=act, glu=True, out_dtype=torch.float32)
        assert relerr(out, ref_nb) < 5e-5


def test_gemm_n_valid(dev):
    from seedx_amd import ops
    dtype = torch.bfloat16
    a, w = rnd((500, 128), dtype, dev, seed=14), rnd((16, 128), dtype, dev, 0.1, seed=15)
    w[4:] = 0
    out = ops.gemm(a, w, out_dtype=torch.float32, n_valid=4)
    assert out.shape == (500, 4)
    assert relerr(out, (a.float() @ w.float().t())[:, :4]) < TOL[torch.float32]


def _conv_ref(x_nhwc, w_flat, bias, stride, upsample):
    B, H, W, Cin = x_nhwc.shape
    Cout = w_flat.shape[0]
    x = x_nhwc.float().permute(0, 3, 1, 2)
    if upsample:
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
    w = w_flat.float().view(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
    y = F.conv2d(x, w, bias, stride=stride, padding=1)
    return y.permute(0, 2, 3, 1).reshape(B, -1, Cout)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,upsample", [(2, 16, 16, 64, 128, 1, False), (1, 10, 12, 128, 64, 1, False),
                                                             (2, 16, 16, 64, 64, 2, False), (2, 8, 8, 64, 128, 1, True),
                                                             (3, 32, 32, 320, 320, 1, False)])
def test_conv3x3(dev, dtype, B, H, W, Cin, Cout, stride, upsample):
    from seedx_amd import ops
    x = rnd((B, H, W, Cin), dtype, dev, seed=16)
    w = rnd((Cout, 9 * Cin), dtype, dev, 0.05, seed=17)
    bias = rnd((Cout,), torch.float32, dev, seed=18)
    ref = _conv_ref(x, w, bias, stride, upsample)
    out = ops.conv3x3(x, w, bias=bias, stride=stride, upsample=upsample, out_dtype=torch.float32)
    assert out.shape == ref.shape and relerr(out, ref) < TOL[torch.float32]
    # time-embedding add + residual
    b2 = rnd((B, Cout), torch.float32, dev, seed=19)
    res = rnd((ref.shape[0] * ref.shape[1], Cout), torch.float32, dev, seed=20)
    out = ops.conv3x3(x, w, bias=bias, bias2d=b2, residual=res, stride=stride, upsample=upsample,
                      out_dtype=torch.float32)
    assert relerr(out, ref + b2[:, None, :] + res.view_as(ref)) < TOL[torch.float32]


@pytest.mark.parametrize("in_dt,out_dt", [(torch.float32, torch.bfloat16), (torch.float32, torch.float32),
                                          (torch.float16, torch.float16), (torch.bfloat16, torch.float32)])
@pytest.mark.parametrize("rows,cols", [(7, 1664), (165, 5120), (512, 640), (3, 4096)])
def test_layernorm(dev, in_dt, out_dt, rows, cols):
    from seedx_amd import ops
    x = rnd((rows, cols), in_dt, dev, 2.0, seed=21) + 0.5
    g, b = rnd((cols,), torch.float32, dev, seed=22), rnd((cols,), torch.float32, dev, seed=23)
    ref = F.layer_norm(x.float(), (cols,), g, b, 1e-6)
    y = ops.layernorm(x, g, b, 1e-6, out_dt)
    assert relerr(y, ref) < max(TOL[out_dt], 1e-5)
    xf = x.float()
    ref = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5) * g
    y = ops.rmsnorm(x, g, 1e-5, out_dt)
    assert relerr(y, ref) < max(TOL[out_dt], 1e-5)


@pytest.mark.parametrize("B,HW,C", [(2, 64, 320), (3, 100, 960), (1, 1024, 1280), (2, 256, 2560), (2, 4096, 640)])
@pytest.mark.parametrize("silu", [False, True])
def test_groupnorm(dev, B, HW, C, silu):
    from seedx_amd import ops
    x = rnd((B, HW, C), torch.float32, dev, 1.5, seed=24) + 0.3
    g, b = rnd((C,), torch.float32, dev, seed=25), rnd((C,), torch.float32, dev, seed=26)
    ref = F.group_norm(x.permute(0, 2, 1), 32, g, b, 1e-5).permute(0, 2, 1)
    if silu:
        ref = F.silu(ref)
    y, raw = ops.groupnorm(x, g, b, 32, 1e-5, silu, torch.float16, want_raw=True)
    assert relerr(y, ref) < TOL[torch.float16]
    assert relerr(raw, x) < TOL[torch.float16]


@pytest.mark.parametrize("B,HW,C1,C2", [(2, 1024, 1280, 1280), (2, 4096, 640, 320), (3, 256, 1280, 640), (1, 4096, 320, 320)])
def test_groupnorm_two_sources_equals_concat(dev, B, HW, C1, C2):
    """sx_groupnorm2 over [x | skip] (UNet up blocks) == GroupNorm of the materialised concatenation, bit for bit the same
    kernel arithmetic (only the addressing differs), and within tolerance of torch's group_norm."""
    from seedx_amd import ops
    x = rnd((B, HW, C1), torch.float32, dev, 1.5, seed=31) + 0.3
    s2 = rnd((B, HW, C2), torch.float32, dev, 0.7, seed=32) - 0.2
    C = C1 + C2
    g, b = rnd((C,), torch.float32, dev, seed=33), rnd((C,), torch.float32, dev, seed=34)
    cat = torch.cat([x, s2], dim=-1).contiguous()
    y1, raw1 = ops.groupnorm(cat, g, b, 32, 1e-5, True, torch.bfloat16, want_raw=True)
    y2, raw2 = ops.groupnorm(x, g, b, 32, 1e-5, True, torch.bfloat16, want_raw=True, x2=s2)
    ref = F.silu(F.group_norm(cat.permute(0, 2, 1), 32, g, b, 1e-5).permute(0, 2, 1))
    assert relerr(y2, ref) < TOL[torch.bfloat16]
    assert torch.equal(raw1, raw2)
    assert relerr(y2, y1) < 1e-3          # identical up to the order of the fp32 partial sums / fp64 atomics


def _attn_ref(q, k, v, scale, causal):
    # q [B,Sq,H,D], k/v [B,Skv,H,D]
    qf, kf, vf = (t.float().permute(0, 2, 1, 3) for t in (q, k, v))
    s = (qf @ kf.transpose(-1, -2)) * scale
    if causal:
        Sq, Skv = s.shape[-2:]
        i = torch.arange(Sq, device=s.device)[:, None]
        j = torch.arange(Skv, device=s.device)[None, :]
        s = s.masked_fill(j > i + (Skv - Sq), float("-inf"))
    o = torch.softmax(s, -1) @ vf
    return o.permute(0, 2, 1, 3).reshape(q.shape[0], q.shape[1], -1)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,H,Sq,Skv,D,causal", [(1, 2, 128, 128, 64, False), (2, 3, 200, 333, 64, False),
                                                  (1, 2, 256, 64, 64, False), (2, 4, 1024, 1024, 104, False),
                                                  (1, 2, 64, 64, 128, False), (1, 5, 165, 165, 128, True),
                                                  (1, 3, 64, 229, 128, True), (1, 2, 300, 300, 64, True),
                                                  (1, 2, 256, 1024, 128, False), (1, 1, 130, 70, 8, False)])
def test_attention_mfma(dev, dtype, B, H, Sq, Skv, D, causal):
    from seedx_amd import ops
    q = rnd((B, Sq, H, D), dtype, dev, seed=27)
    k = rnd((B, Skv, H, D), dtype, dev, seed=28)
    v = rnd((B, Skv, H, D), dtype, dev, seed=29)
    scale = 1.0 / math.sqrt(D)
    ref = _attn_ref(q, k, v, scale, causal)
    out = ops.attention(q, k, v, scale, causal)
    # P is rounded to 16 bit before P·V: error ~ eps/2 relative
    assert relerr(out, ref) < (1.2e-3 if dtype == torch.float16 else 8e-3)


def test_attention_fused_qkv_layouts(dev):
    """ViT per-head-interleaved [T, H, 3, D] and blocked [T, 3, H, D] fused-QKV layouts through strided views."""
    from seedx_amd import ops
    dtype, B, S, H, D = torch.float16, 2, 160, 4, 104
    qkv = rnd((B, S, H, 3, D), dtype, dev, seed=30)
    q, k, v = qkv[:, :, :, 0], qkv[:, :, :, 1], qkv[:, :, :, 2]
    out = ops.attention(q, k, v, D ** -0.5)
    assert relerr(out, _attn_ref(q, k, v, D ** -0.5, False)) < 1.2e-3
    qkv = rnd((B, S, 3, H, 64), dtype, dev, seed=31)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    out = ops.attention(q, k, v, 0.125)
    assert relerr(out, _attn_ref(q, k, v, 0.125, False)) < 1.2e-3


def test_attention_spiked_scores(dev):
    """A large outlier key late in the sequence forces the online-softmax rescale path."""
    from seedx_amd import ops
    dtype, B, H, S, D = torch.float16, 1, 2, 512, 64
    q, k, v = rnd((B, S, H, D), dtype, dev, seed=32), rnd((B, S, H, D), dtype, dev, seed=33), rnd((B, S, H, D), dtype, dev, seed=34)
    k[:, 400] = q[:, 7] * 3.0
    out = ops.attention(q, k, v, 0.125)
    assert relerr(out, _attn_ref(q, k, v, 0.125, False)) < 1.2e-3


@pytest.mark.parametrize("B,H,Sq,Skv,D", [(2, 32, 64, 256, 160), (1, 16, 1, 65, 64), (2, 16, 64, 320, 64), (1, 32, 64, 64, 128)])
def test_attention_small(dev, B, H, Sq, Skv, D):
    from seedx_amd import ops
    dtype = torch.float16
    q, k, v = rnd((B, Sq, H, D), dtype, dev, seed=35), rnd((B, Skv, H, D), dtype, dev, seed=36), rnd((B, Skv, H, D), dtype, dev, seed=37)
    out = ops.attention_small(q, k, v, D ** -0.5)
    assert relerr(out, _attn_ref(q, k, v, D ** -0.5, False)) < TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(1, 512, 1024), (1, 5120, 5120), (3, 130, 256), (8, 1000, 512)])
def test_gemv(dev, dtype, M, N, K):
    from seedx_amd import ops
    x, w = rnd((M, K), dtype, dev, seed=38), rnd((N, K), dtype, dev, 0.05, seed=39)
    res = rnd((M, N), torch.float32, dev, seed=40)
    out = ops.gemv(x, w, residual=res, out_dtype=torch.float32)
    assert relerr(out, x.float() @ w.float().t() + res) < 5e-5
    out = ops.gemv(x, w, act="silu")
    assert relerr(out, F.silu(x.float() @ w.float().t())) < TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(2, 512, 256), (8, 5120, 5120), (5, 8192, 1728), (16, 640, 832), (8, 32384, 512)])
def test_gemv_mfma_rows(dev, dtype, M, N, K):
    """2..16 activation rows take the MFMA skinny-GEMM path (R = 1 and R = 2 row groups, ragged k split); it must agree
    with the fp32 reference and with the VALU path on the same inputs."""
    from seedx_amd import _lib, ops
    x, w = rnd((M, K), dtype, dev, seed=38), rnd((N, K), dtype, dev, 0.05, seed=39)
    res = rnd((M, N), torch.float32, dev, seed=40)
    ref = x.float() @ w.float().t()
    lib = _lib.load()
    lib.sx_gemv_force_valu(2)                         # MFMA path for every M >= 2 (auto picks it from M = 5)
    try:
        out = ops.gemv(x, w, residual=res, out_dtype=torch.float32)
        assert relerr(out, ref + res) < 5e-5
        assert relerr(ops.gemv(x, w, act="silu"), F.silu(ref)) < TOL[dtype]
        if M <= 8:
            lib.sx_gemv_force_valu(1)
            valu = ops.gemv(x, w, residual=res, out_dtype=torch.float32)
            assert relerr(out, valu) < 2e-5
            lib.sx_gemv_force_valu(2)
        # GLU: packed [16 linear | 16 gate] groups, out = lin * silu(gate)
        I = N // 2
        lin, gate = w[:I], w[I:2 * I]
        packed = torch.cat([lin.view(I // 16, 16, K), gate.view(I // 16, 16, K)], dim=1).reshape(2 * I, K).contiguous()
        g = ops.gemv(x, packed, act="silu", glu=True, out_dtype=torch.float32)
        assert relerr(g, (x.float() @ lin.float().t()) * F.silu(x.float() @ gate.float().t())) < 5e-5
    finally:
        lib.sx_gemv_force_valu(0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(5, 512, 256), (16, 5120, 5120), (8, 8192, 1728), (16, 32384, 512)])
def test_gemv_decode_tile_layout(dev, dtype, M, N, K):
    """The decode-tile weight layout [N/16][K/32][16][32] (ops.pack_decode_tiles) must give bit-identical results to the
    row-major weight on the MFMA skinny path: plain, fp32 residual epilogue, and GLU-packed rows."""
    from seedx_amd import ops
    x, w = rnd((M, K), dtype, dev, seed=44), rnd((N, K), dtype, dev, 0.05, seed=45)
    res = rnd((M, N), torch.float32, dev, seed=46)
    t = ops.pack_decode_tiles(w)
    assert t.shape == w.shape and not torch.equal(t, w)
    # element (n, k) of the row-major weight sits at tile (n // 16, k // 32), row n % 16, column k % 32
    assert torch.equal(t.view(N // 16, K // 32, 16, 32)[3, 1, 5], w[3 * 16 + 5, 32:64])
    assert torch.equal(ops.gemv(x, w), ops.gemv(x, w, w_tiles=t))
    a = ops.gemv(x, w, residual=res, out_dtype=torch.float32)
    assert torch.equal(a, ops.gemv(x, w, residual=res, out_dtype=torch.float32, w_tiles=t))
    assert relerr(a, x.float() @ w.float().t() + res) < 5e-5
    assert torch.equal(ops.gemv(x, w, act="silu", glu=True), ops.gemv(x, w, act="silu", glu=True, w_tiles=t))
    # fewer than 5 rows keep the VALU path and the row-major weight (w_tiles is ignored)
    assert torch.equal(ops.gemv(x[:2].contiguous(), w), ops.gemv(x[:2].contiguous(), w, w_tiles=t))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(5, 512, 320), (16, 5120, 5120), (16, 5120, 13824), (9, 1024, 8192 + 64), (16, 2048, 704)])
def test_gemv_tiled_activations_and_split_k(dev, dtype, M, N, K):
    """Decode-step activations as operand tiles (ops.Tiled16 / SX_TILED16): the three producers (RMSNorm, skinny-GEMM epilogue
    incl. GLU, decode-attention combine is covered in test_attn_decode_b_tiled) must place element (m, k) at tile k // 32, row m,
    column k % 32, and the skinny GEMM must give the SAME BITS from a tiled x as from the row-major x. Split-K over workgroups
    (workspace given, K >= 8192): deterministic (20 launches), equal to the unsplit result up to fp32 re-association, counters
    left at zero. K / 64 not divisible by the 4 waves x 4 k-steps exercises the peeled partial rounds — the guards there must
    be scalar branches: MFMA ignores EXEC (a VGPR-derived guard multiplied never-loaded registers in)."""
    from seedx_amd import ops
    x32 = rnd((M, K), torch.float32, dev, seed=51)
    gamma = rnd((K,), torch.float32, dev, seed=52)
    w = rnd((N, K), dtype, dev, 0.05, seed=53)
    t = ops.pack_decode_tiles(w)
    res = rnd((M, N), torch.float32, dev, seed=54)
    # producer 1: RMSNorm
    if K <= 6144:                      # the norm kernel's register-resident limit
        h = ops.rmsnorm(x32, gamma, 1e-5, dtype)
        ht = ops.rmsnorm(x32, gamma, 1e-5, dtype, tiled=True)
        assert ht.t.shape == (1, K // 32, 16, 32) and torch.equal(ht.dense(), h)
        assert torch.equal(ht.t[0, 3, M - 1], h[M - 1, 96:128])
    else:                              # tile by hand (what the GLU epilogue produces for the down projection)
        h = x32.to(dtype)
        ht = ops.Tiled16(M, K, dtype, dev)
        ht.t.fill_(float("nan"))       # padding rows may hold anything
        ht.t[0, :, :M] = h.view(M, K // 32, 32).permute(1, 0, 2)
    # consumer: same bits from tiled and row-major x
    a = ops.gemv(h, w, residual=res, out_dtype=torch.float32, w_tiles=t)
    b = ops.gemv(ht, w, residual=res, out_dtype=torch.float32, w_tiles=t)
    assert torch.equal(a, b)
    assert relerr(a, h.float() @ w.float().t() + res) < 5e-5
    # producer 2: skinny-GEMM epilogue, plain and GLU
    y, yt = ops.gemv(ht, w, w_tiles=t), ops.gemv(ht, w, w_tiles=t, y_tiled=True)
    assert yt.t.shape == (1, N // 32, 16, 32) and torch.equal(yt.dense(), y)
    g, gt = ops.gemv(ht, w, act="silu", glu=True, w_tiles=t), ops.gemv(ht, w, act="silu", glu=True, w_tiles=t, y_tiled=True)
    if (N // 2) % 32 == 0:
        assert torch.equal(gt.dense(), g)
    # split-K
    ws = torch.zeros(16384 + 8 * 16 * N * 4, dtype=torch.uint8, device=dev)
    c = ops.gemv(ht, w, residual=res, out_dtype=torch.float32, w_tiles=t, workspace=ws)
    for _ in range(20):
        assert torch.equal(c, ops.gemv(ht, w, residual=res, out_dtype=torch.float32, w_tiles=t, workspace=ws))
    assert relerr(c, a) < 2e-6
    assert int(ws[:16384].view(torch.int32).abs().sum()) == 0, "arrival counters must be left at zero"
    if K >= 8192 and N // 16 < 512:
        assert int(ws[16384:].view(torch.int32).abs().sum()) != 0, "this shape was expected to split K"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("nsplit", [8, 1, 3])
@pytest.mark.parametrize("G,H,D,tiled", [(7, 8, 128, True), (1, 5, 128, False), (16, 4, 64, True), (3, 2, 112, False), (20, 4, 128, True)])
def test_attn_decode_fused_equals_three_kernel_form(dev, dtype, G, H, D, tiled, nsplit):
    """sx_attn_decode_fused (RoPE + KV append + split-KV attention + combine in one launch) against sx_rope_kv_append_b +
    sx_attn_decode_b: the SAME bits in the output and in both caches, for positions 0, mid-chunk, chunk boundaries and the last
    cache slot; positions outside the cache write nothing. The arrival counters are left at zero. nsplit = 1 is the
    no-partials fast path the lock-step batch uses (one workgroup per head and sequence)."""
    from seedx_amd import ops
    T = 96
    pos_list = ([0, 5, 95, 40, 17, 64, 33, 8, 15, 16, 31, 32, 47, 48, 63, 94] * 2)[:G]
    pos = torch.tensor(pos_list, dtype=torch.int32, device=dev)
    qkv = rnd((G, 3 * H * D), dtype, dev, seed=71)
    kc0, vc0 = rnd((G, H, T, D), dtype, dev, seed=72), rnd((G, H, T, D), dtype, dev, seed=73)
    inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2).float() / D))
    fr = torch.outer(torch.arange(T).float(), inv)
    cos, sin = fr.cos().to(dev).contiguous(), fr.sin().to(dev).contiguous()
    scale = 1.0 / D ** 0.5
    # three-kernel form
    q1, kc1, vc1 = qkv.clone(), kc0.clone(), vc0.clone()
    ops.rope_kv_append_b(q1, kc1, vc1, cos, sin, pos, G, 1, H, D)
    ref = ops.attn_decode_b(q1[:, :H * D].unflatten(1, (H, D)), kc1, vc1, pos + 1, scale, nsplit=nsplit)
    # fused
    q2, kc2, vc2 = qkv.clone(), kc0.clone(), vc0.clone()
    cnt = torch.zeros(G * H, dtype=torch.int32, device=dev)
    for rep in range(3):               # repeated launches: counters must have been left at zero
        out = ops.attn_decode_fused(q2, kc2, vc2, pos, cos, sin, scale, H, D, cnt, nsplit=nsplit, out_tiled=tiled and (H * D) % 32 == 0)
        got = out.dense() if isinstance(out, ops.Tiled16) else out
        assert torch.equal(got, ref), f"rep {rep}: max diff {(got.float() - ref.float()).abs().max().item()}"
    assert torch.equal(kc2, kc1) and torch.equal(vc2, vc1) and torch.equal(q2, qkv)
    assert int(cnt.abs().sum()) == 0
    # a position past the cache: nothing is written
    bad = pos.clone(); bad[0] = T
    kc3, vc3 = kc0.clone(), vc0.clone()
    ops.attn_decode_fused(qkv.clone(), kc3, vc3, bad, cos, sin, scale, H, D, cnt, nsplit=nsplit)
    assert torch.equal(kc3[0], kc0[0]) and torch.equal(vc3[0], vc0[0])


def test_attn_decode_b_tiled(dev):
    from seedx_amd import ops
    H, D, T = 8, 128, 96
    dt = torch.bfloat16
    for G in (7, 20):                                  # 20: sequences 16..19 land in the second block of tiles
        q = rnd((G, H, D), dt, dev, seed=61)
        kc, vc = rnd((G, H, T, D), dt, dev, seed=62), rnd((G, H, T, D), dt, dev, seed=63)
        ctx = torch.tensor(([1, 5, 96, 40, 17, 64, 33] * 3)[:G], dtype=torch.int32, device=dev)
        a = ops.attn_decode_b(q, kc, vc, ctx, 0.088)
        b = ops.attn_decode_b(q, kc, vc, ctx, 0.088, out_tiled=True)
        assert b.t.shape == ((G + 15) // 16, H * D // 32, 16, 32) and torch.equal(b.dense(), a)
    G = 7


@pytest.mark.parametrize("ctx", [1, 17, 166, 1000])
def test_attn_decode(dev, ctx):
    from seedx_amd import ops
    dtype, H, D, Tmax = torch.bfloat16, 8, 128, 1024
    q = rnd((H, D), dtype, dev, seed=41)
    kc, vc = rnd((H, Tmax, D), dtype, dev, seed=42), rnd((H, Tmax, D), dtype, dev, seed=43)
    ctx_dev = torch.tensor([ctx], dtype=torch.int32, device=dev)
    out = ops.attn_decode(q, kc, vc, ctx_dev, D ** -0.5, nsplit=4)
    s = torch.einsum("hd,htd->ht", q.float(), kc[:, :ctx].float()) * D ** -0.5
    ref = torch.einsum("ht,htd->hd", torch.softmax(s, -1), vc[:, :ctx].float()).reshape(1, -1)
    assert relerr(out, ref) < TOL[dtype]


def test_rope_kv_append(dev):
    from seedx_amd import ops
    dtype, T, H, D, Tmax, pos0 = torch.float16, 5, 3, 128, 64, 7
    qkv = rnd((T, 3 * H * D), dtype, dev, seed=44)
    inv = 1.0 / (10000 ** (torch.arange(0, D, 2).float() / D))
    fr = torch.outer(torch.arange(Tmax).float(), inv)
    cos_t, sin_t = fr.cos().to(dev).contiguous(), fr.sin().to(dev).contiguous()
    kc = torch.zeros((H, Tmax, D), dtype=dtype, device=dev)
    vc = torch.zeros_like(kc)
    q0 = qkv.clone()
    ops.rope_kv_append(qkv, kc, vc, cos_t, sin_t, torch.tensor([pos0], dtype=torch.int32, device=dev), H, D)

    def rope(x, pos):  # x [T,H,D]
        c = torch.cat([cos_t, cos_t], -1)[pos].to(dtype).float()[:, None, :]
        s = torch.cat([sin_t, sin_t], -1)[pos].to(dtype).float()[:, None, :]
        xf = x.float()
        rot = torch.cat([-xf[..., D // 2:], xf[..., :D // 2]], -1)
        return xf * c + rot * s
    pos = torch.arange(pos0, pos0 + T, device=dev)
    v3 = q0.view(T, 3, H, D)
    assert relerr(qkv.view(T, 3, H, D)[:, 0], rope(v3[:, 0], pos)) < TOL[dtype]
    assert relerr(kc[:, pos0:pos0 + T].permute(1, 0, 2), rope(v3[:, 1], pos)) < TOL[dtype]
    assert torch.equal(vc[:, pos0:pos0 + T].permute(1, 0, 2), v3[:, 2])
    assert kc[:, :pos0].abs().sum() == 0 and kc[:, pos0 + T:].abs().sum() == 0


def test_embedding_scatter_greedy(dev):
    from seedx_amd import ops
    table = rnd((100, 64), torch.float16, dev, seed=45)
    ids = torch.tensor([3, 99, 0, 3], dtype=torch.int32, device=dev)
    e = ops.embedding(ids, table)
    assert torch.equal(e, table[ids.long()].float())
    dst = torch.zeros((10, 64), dtype=torch.float32, device=dev)
    ops.scatter_rows(e[:2].contiguous(), torch.tensor([7, 2], dtype=torch.int32, device=dev), dst)
    assert torch.equal(dst[7], e[0]) and torch.equal(dst[2], e[1]) and dst[0].abs().sum() == 0
    # greedy + logits rule. img ids = [50 (<img>), 51..54, 55 (</img>)]
    vocab = 90
    img = torch.tensor([50, 51, 52, 53, 54, 55], dtype=torch.int32, device=dev)
    logits = rnd((1, 96), torch.float32, dev, seed=46) - 5.0   # all negative → zeroed image ids would win
    logits[0, 90:] = 100.0                                       # padding columns beyond vocab must be ignored
    cur = torch.tensor([52], dtype=torch.int32, device=dev)
    outs = torch.full((4,), -1, dtype=torch.int32, device=dev)
    step = torch.tensor([1], dtype=torch.int32, device=dev)
    ops.greedy_next(logits.clone(), vocab, img, cur, cur, outs, step)
    assert cur.item() == 53 and outs[1].item() == 53                # forced chain
    cur.fill_(7)
    l2 = logits.clone()
    ops.greedy_next(l2, vocab, img, cur, cur)
    # reference semantics: image ids (except <img>) set to 0.0, then argmax → first zeroed id wins here
    ref = logits[0, :vocab].clone()
    ref[img[1:].long()] = 0.0
    assert cur.item() == int(ref.argmax()) == 51
    assert torch.equal(l2[0, :vocab], ref)
    cur.fill_(55)                                                   # </img> is NOT in img_ids[:-1] → normal rule
    l3 = logits.clone(); l3[0, 10] = 3.0
    ops.greedy_next(l3, vocab, img, cur, cur)
    assert cur.item() == 10


def test_elementwise(dev):
    from seedx_amd import ops
    x = rnd((3, 1001), torch.float32, dev, seed=47)
    assert torch.equal(ops.cast(x, torch.float16), x.half()) and torch.equal(ops.cast(x, torch.bfloat16), x.bfloat16())
    assert torch.equal(ops.cast(x.half(), torch.float32), x.half().float())
    a, b = rnd((50, 64), torch.float32, dev, seed=48), rnd((50, 64), torch.float32, dev, seed=49)
    assert torch.equal(ops.add(a, b), a + b)
    dst = torch.zeros((50, 192), dtype=torch.float32, device=dev)
    ops.copy2d(a, dst, 0); ops.copy2d(b, dst, 128)
    assert torch.equal(dst[:, :64], a) and torch.equal(dst[:, 128:], b) and dst[:, 64:128].abs().sum() == 0
    # patchify == conv1 unfold
    img = rnd((2, 3, 56, 56), torch.float32, dev, seed=50)
    p = ops.patchify(img, 14, 640, torch.float16)
    ref = F.unfold(img, 14, stride=14).transpose(1, 2).reshape(-1, 588)
    assert torch.equal(p[:, :588], ref.half()) and p[:, 588:].abs().sum() == 0
    # im2col small
    xs = rnd((2, 6, 5, 4), torch.float32, dev, seed=51)
    col = ops.im2col3x3_small(xs, 64, torch.float16)
    w = rnd((16, 36), torch.float16, dev, 0.2, seed=52)
    wp = torch.zeros((16, 64), dtype=torch.float16, device=dev); wp[:, :36] = w
    y = ops.gemm(col, wp, out_dtype=torch.float32)
    refc = _conv_ref(xs.half(), w, None, 1, False).reshape(-1, 16)
    assert relerr(y, refc) < TOL[torch.float32]
    # avgpool tokens
    t = rnd((2, 256, 96), torch.float32, dev, seed=53)
    assert relerr(ops.avgpool_tokens(t, 4), F.avg_pool1d(t.permute(0, 2, 1), 4, 4).permute(0, 2, 1)) < 1e-6
    # timestep embedding (flip_sin_to_cos, shift 0)
    ts = torch.tensor([981.0, 1.0, 1024.0], device=dev)
    emb = ops.timestep_embedding(ts, 320, torch.float32)
    half = 160
    freqs = torch.exp(-math.log(10000) * torch.arange(half, device=dev).float() / half)
    arg = ts[:, None] * freqs[None]
    assert (emb - torch.cat([arg.cos(), arg.sin()], -1)).abs().max() < 2e-4
    idx = torch.tensor([1], dtype=torch.int32, device=dev)
    emb2 = ops.timestep_embedding(ts, 320, torch.float32, idx_dev=idx, n=2)
    assert torch.equal(emb2[0], emb[1]) and torch.equal(emb2[1], emb[1])
    # layouts
    lat = rnd((2, 4, 8, 8), torch.float32, dev, seed=54)
    nhwc = ops.nchw_to_nhwc(lat, ld=8)
    assert torch.equal(nhwc[..., :4], lat.permute(0, 2, 3, 1).reshape(2, 64, 4)) and nhwc[..., 4:].abs().sum() == 0
    assert torch.equal(ops.nhwc_to_nchw(nhwc, 4, 8, 8), lat)
    c = torch.tensor([5], dtype=torch.int32, device=dev); ops.add_i32(c, 3); assert c.item() == 8


@pytest.mark.parametrize("mode", [0, 1])
def test_cfg_euler_step(dev, mode):
    from seedx_amd import ops
    HW, Cl = 64, 4
    nb = 2 if mode == 0 else 3
    ld = 4 if mode == 0 else 8
    lat = rnd((1, HW, Cl), torch.float32, dev, 5.0, seed=55)
    eps = rnd((nb, HW, Cl), torch.float32, dev, seed=56)
    sig = torch.tensor([14.6, 12.1, 9.7, 0.0], device=dev)
    step = torch.tensor([1], dtype=torch.int32, device=dev)
    scaled = torch.zeros((nb, HW, ld), dtype=torch.float32, device=dev)
    lat0 = lat.clone()
    ops.cfg_euler_step(eps, lat, scaled, sig, step, nb, Cl, ld, 7.5, 1.5, mode)
    s, sn = 12.1, 9.7
    if mode == 0:
        e = eps[0] + 7.5 * (eps[1] - eps[0])
    else:
        x0 = [lat0[0] - s * eps[i] for i in range(3)]
        x0c = x0[2] + 7.5 * (x0[0] - x0[1]) + 1.5 * (x0[1] - x0[2])
        e = (x0c - lat0[0]) / (-s)
    ref = lat0[0] + e * (sn - s)
    assert relerr(lat[0], ref) < 1e-5
    for kk in range(nb):
        assert relerr(scaled[kk, :, :Cl], ref / math.sqrt(sn * sn + 1)) < 1e-5
    assert scaled[..., Cl:].abs().sum() == 0


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_gemm_fused_groupnorm_statistics(dev, dtype):
    """sx_gemm_gn: the statistics pass of the GroupNorm that reads a GEMM's / conv's fp32 output, accumulated by that launch's
    epilogue (ping-pong tiles). Checked against fp64 sums of the stored output; C = 320 puts group boundaries (10 channels) inside
    the 4-column pieces a lane holds; the consumer's apply pass on the fused statistics must equal the unfused GroupNorm."""
    from seedx_amd import ops
    g = torch.Generator().manual_seed(11)
    G = 32
    # linear + bias + fp32 residual (proj_out / attention out-proj epilogue), 2 samples x 1024 rows, N = 640 and 1280
    from seedx_amd import _lib
    lib = _lib.load()
    for N, K, B, HW in ((640, 640, 32, 4096), (1280, 1280, 32, 1024)):       # the UNet's out-proj shapes at 64^2 / 32^2, 32 samples
        assert lib.sx_gemm_pick_tile(B * HW, N, K, 0, 0) in (7, 8), "test shape must run on a ping-pong tile"
        a = (torch.randn(B * HW, K, generator=g) * 0.5).to(dtype).to(dev)
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dtype).to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        res = torch.randn(B * HW, N, generator=g).to(dev)
        arena = ops.GnStats.arena(1, B, G, dev)
        gs = ops.GnStats(arena[0], G, HW)
        out = ops.gemm(a, w, bias=bias, residual=res, out_dtype=torch.float32, gn=gs)
        plain = ops.gemm(a, w, bias=bias, residual=res, out_dtype=torch.float32)
        assert torch.equal(out, plain), "the fused statistics must not change the stored output"
        assert gs.ready
        o64 = out.double().view(B, HW, G, N // G)
        ref = torch.stack([o64.sum(dim=(1, 3)), (o64 * o64).sum(dim=(1, 3))], dim=-1)
        assert torch.allclose(gs.buf, ref, rtol=2e-6, atol=1e-3), (gs.buf - ref).abs().max()
        gamma, beta = torch.randn(N, generator=g).to(dev), torch.randn(N, generator=g).to(dev)
        y_f = ops.groupnorm(out.view(B, HW, N), gamma, beta, G, 1e-5, True, dtype, stats=gs)
        y_u = ops.groupnorm(out.view(B, HW, N), gamma, beta, G, 1e-5, True, dtype)
        assert (y_f.float() - y_u.float()).abs().max() <= 2e-2 * y_u.float().abs().max() * (1 if dtype == torch.bfloat16 else 0.1)
    # 3x3 conv 320 -> 320 at 32 x 32 (resnet conv1 with the per-sample time add): 10 channels per group
    B, H, Cin, Co = 8, 128, 320, 320
    assert lib.sx_gemm_pick_tile(B * H * H, Co, 9 * Cin, 0, 1) in (7, 8)
    x = (torch.randn(B, H, H, Cin, generator=g) * 0.5).to(dtype).to(dev)
    w = (torch.randn(Co, 9 * Cin, generator=g) / (9 * Cin) ** 0.5).to(dtype).to(dev)
    b2 = torch.randn(B, Co, generator=g).to(dev)
    arena = ops.GnStats.arena(1, B, G, dev)
    gs = ops.GnStats(arena[0], G, H * H)
    out = ops.conv3x3(x, w, bias=torch.zeros(Co, device=dev), bias2d=b2, out_dtype=torch.float32, gn=gs)
    assert gs.ready
    o64 = out.double().view(B, H * H, G, Co // G)
    ref = torch.stack([o64.sum(dim=(1, 3)), (o64 * o64).sum(dim=(1, 3))], dim=-1)
    assert torch.allclose(gs.buf, ref, rtol=2e-6, atol=1e-3), (gs.buf - ref).abs().max()
    # a launch that does not run on a ping-pong tile reports ready = False and the consumer falls back to its own pass
    a = torch.randn(256, 64, generator=g).to(dtype).to(dev)
    w = torch.randn(64, 64, generator=g).to(dtype).to(dev)
    gs = ops.GnStats(ops.GnStats.arena(1, 1, G, dev)[0], G, 256)
    out = ops.gemm(a, w, out_dtype=torch.float32, gn=gs)
    assert not gs.ready and float(gs.buf.abs().sum()) == 0.0
    y = ops.groupnorm(out.view(1, 256, 64), torch.ones(64, device=dev), torch.zeros(64, device=dev), G, 1e-5, False, dtype, stats=gs)
    y2 = ops.groupnorm(out.view(1, 256, 64), torch.ones(64, device=dev), torch.zeros(64, device=dev), G, 1e-5, False, dtype)
    assert torch.equal(y, y2)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M", [5, 16, 32])
def test_gemv_rmsnorm_fold(dev, dtype, M):
    """RMSNorm folded into the decode step's skinny GEMMs (sx_gemv_args.x16_out / row_ssq_*): the residual GEMV (o / down shapes,
    incl. the split-K one) emits the new residual stream x also as 16-bit operand tiles and its rows' sums of squares per
    workgroup; the GEMV behind the norm, with gamma folded into its weights, then equals RMSNorm(x) @ W^T — compared with the
    unfused kernels (norm launch + plain weights) and with fp32 torch."""
    from seedx_amd import ops
    g = torch.Generator().manual_seed(23 + M)
    H, eps = 5120, 1e-5
    for K in (5120, 13824):                                              # o-proj, down-proj (split-K over workgroups)
        a = (torch.randn(M, K, generator=g) * 0.5).to(dtype).to(dev)
        w = (torch.randn(H, K, generator=g) / K ** 0.5).to(dtype).to(dev)
        res = torch.randn(M, H, generator=g).to(dev)
        ws = torch.zeros(16384 + 8 * 32 * H * 4, dtype=torch.uint8, device=dev)
        wt = ops.pack_decode_tiles(w)
        at = ops.Tiled16(M, K, dtype, dev)                                 # the operand as MFMA tiles [K/32][16][32]
        nb = (M + 15) // 16                                                # M = 32: two 16-row blocks of tiles
        pad = torch.zeros(16 * nb, K, dtype=dtype, device=dev)
        pad[:M] = a
        at.t.copy_(pad.view(nb, 16, K // 32, 32).permute(0, 2, 1, 3))
        y_plain = ops.gemv(at, w, residual=res, out_dtype=torch.float32, w_tiles=wt, workspace=ws)
        y, x16, ssq = ops.gemv(at, w, residual=res, out_dtype=torch.float32, w_tiles=wt, workspace=ws, emit_norm=True)
        assert torch.equal(y, y_plain), "emitting the norm inputs must not change the fp32 output"
        assert torch.equal(x16.dense(), y.to(dtype)), "x16 = the fp32 output rounded once to 16 bits, as operand tiles"
        ref_ssq = (y.double() ** 2).sum(dim=1)
        got = ssq.double().sum(dim=1)[:M]
        assert torch.allclose(got, ref_ssq, rtol=1e-5), (got, ref_ssq)
        # the same launch over 20-row decode tiles (w_layout 2: 256 equal workgroups for N = 5120) gives the same bits
        w20 = ops.pack_decode_tiles20(w)
        y20, x16b, ssq20 = ops.gemv(at, w, residual=res, out_dtype=torch.float32, w_tiles20=w20, workspace=ws, emit_norm=True)
        assert ssq20.shape == (16 * nb, 256) and torch.equal(x16b.dense(), y20.to(dtype))
        # K = 5120: same per-wave k ranges and MFMA order → the same bits; K = 13824: the 16-row path splits K over 4 workgroups
        # (another summation order), the balanced path does not
        assert torch.equal(y20, y_plain) if K == 5120 else relerr(y20, y_plain) < 2e-6
        assert torch.allclose(ssq20.double().sum(dim=1)[:M], (y20.double() ** 2).sum(dim=1), rtol=1e-5)
        assert torch.equal(ops.gemv(at, w, residual=res, out_dtype=torch.float32, w_tiles20=w20), y20)
        # consumer: qkv-like (plain) and gate|up-like (SiLU-GLU) projections behind the norm
        gamma = (1.0 + 0.1 * torch.randn(H, generator=g)).to(dev)
        for N, glu in ((15360, False), (2 * 13824, True)):
            wn = (torch.randn(N, H, generator=g) / H ** 0.5).to(dev)
            if glu:
                from seedx_amd.llama import glu_pack_rows
                pack = lambda t: glu_pack_rows(t[:N // 2].contiguous(), t[N // 2:].contiguous())
            else:
                pack = lambda t: t
            w_plain = pack(wn.to(dtype))
            w_fold = pack((wn * gamma[None, :]).to(dtype))
            kw = dict(act="silu", glu=True) if glu else {}
            out_f = ops.gemv(x16, w_fold, w_tiles=ops.pack_decode_tiles(w_fold), ssq_in=(ssq, H, eps), **kw)
            h = ops.rmsnorm(y, gamma, eps, dtype, tiled=True)
            out_u = ops.gemv(h, w_plain, w_tiles=ops.pack_decode_tiles(w_plain), **kw)
            hn = y * torch.rsqrt((y * y).mean(dim=1, keepdim=True) + eps) * gamma
            z = hn @ wn.t()
            ref = torch.nn.functional.silu(z[:, N // 2:]) * z[:, :N // 2] if glu else z
            e_f, e_u = relerr(out_f, ref), relerr(out_u, ref)
            print(f"rmsnorm fold {dtype} M={M} K={K} N={N} glu={glu}: folded {e_f:.2e}, unfused {e_u:.2e} vs fp32")
            tol = 2e-3 if dtype == torch.float16 else 1.6e-2
            assert e_f < tol and e_f < 1.5 * e_u + 1e-4


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_gemm_layernorm_fold(dev, dtype):
    """sx_gemm_ln: a LayerNorm folded into the GEMMs either side of it (the SDXL transformer blocks' norm1/2/3, [ext
    BasicTransformerBlock]). Producer = out-projection + fp32 residual that also stores the 16-bit copy of its output and the rows'
    (sum, sum of squares); consumer = the projection behind the norm on that copy with gamma folded into the weight, applying
    (mu, rstd) per row in its epilogue. Checked: producer output bit-equal to the plain launch, copy = rounded output, sums vs
    fp64; consumer against LayerNorm + projection in fp64, beside the unfused launches (sx_layernorm + sx_gemm) on the same data."""
    from seedx_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(23)
    for M, C in ((16384, 1280), (32768, 640)):
        consumers = [(3 * C, False, None), (C, False, None), (8 * C, True, "gelu")]
        assert ops.ln_fold_ok(M, C, [(n, glu) for n, glu, _ in consumers], [C, 4 * C])
        # producer: out-projection of M rows + bias + fp32 residual with a per-row mean offset (|mu| ~ sigma / 2) and outlier channels
        a = (torch.randn(M, C, generator=g) * 0.5).to(dtype).to(dev)
        wo = (torch.randn(C, C, generator=g) / C ** 0.5).to(dtype).to(dev)
        bo = torch.randn(C, generator=g).to(dev)
        res = (torch.randn(M, C, generator=g) * 2.0 + torch.randn(M, 1, generator=g)).to(dev)
        res[:, :3] *= 8.0
        rows = ops.LnRows(M, C, dtype, dev)
        hs = ops.gemm(a, wo, bias=bo, residual=res, out_dtype=torch.float32, ln_emit=rows)
        plain = ops.gemm(a, wo, bias=bo, residual=res, out_dtype=torch.float32)
        assert torch.equal(hs, plain), "the producer role must not change the stored fp32 output"
        assert torch.equal(rows.x16, hs.to(dtype)), "x16 = the stored output rounded to the operand dtype"
        h64 = hs.double()
        ref_stats = torch.stack([h64.sum(1), (h64 * h64).sum(1)], dim=1)
        assert torch.allclose(rows.stats, ref_stats, rtol=3e-6, atol=1e-3), (rows.stats - ref_stats).abs().max()
        # consumers
        gamma = (1 + 0.2 * torch.randn(C, generator=g)).to(dev)
        beta = (0.2 * torch.randn(C, generator=g)).to(dev)
        mu, var = h64.mean(1, keepdim=True), h64.var(1, unbiased=False, keepdim=True)
        ln64 = (h64 - mu) / (var + 1e-5).sqrt() * gamma.double() + beta.double()
        n16 = ops.layernorm(hs, gamma, beta, 1e-5, dtype)
        for N, glu, act in consumers:
            w = (torch.randn(N, C, generator=g) / C ** 0.5).to(dtype).to(dev)
            b = torch.randn(N, generator=g).to(dev) if glu else None
            wf, cs, bf = ops.fold_layernorm(w, b, gamma, beta)
            y_f = ops.gemm(rows.x16, wf, bias=bf, act=act, glu=glu, ln_apply=(rows, cs, 1e-5))
            y_u = ops.gemm(n16, w, bias=b, act=act, glu=glu)
            z = ln64 @ w.double().T + (b.double() if b is not None else 0.0)
            if glu:      # GLU-packed rows: 16 linear | 16 gate interleaved — compare through the same unpacked reference
                zz = z.view(M, N // 32, 2, 16)
                z = (zz[:, :, 0] * torch.nn.functional.gelu(zz[:, :, 1])).reshape(M, N // 2)
            e_f, e_u = relerr(y_f, z.float()), relerr(y_u, z.float())
            bound = 1.2e-2 if dtype == torch.bfloat16 else 1.0e-3
            assert e_f < bound and e_f < 1.35 * e_u + 1e-5, (M, C, N, glu, e_f, e_u)
    # the 256 x 256 tile's producer / plain consumer (the UNet shapes above pick 256 x 320 for them): forced
    lib.sx_gemm_force_tile(7)
    try:
        M, C, N = 4096 + 40, 1280, 1280                     # a ragged last row tile on the way
        a = (torch.randn(M, C, generator=g) * 0.5).to(dtype).to(dev)
        wo = (torch.randn(C, C, generator=g) / C ** 0.5).to(dtype).to(dev)
        res = (torch.randn(M, C, generator=g) * 2.0 + 1.0).to(dev)
        rows = ops.LnRows(M, C, dtype, dev)
        hs = ops.gemm(a, wo, residual=res, out_dtype=torch.float32, ln_emit=rows)
        assert torch.equal(hs, ops.gemm(a, wo, residual=res, out_dtype=torch.float32)) and torch.equal(rows.x16, hs.to(dtype))
        h64 = hs.double()
        ref_stats = torch.stack([h64.sum(1), (h64 * h64).sum(1)], dim=1)
        assert torch.allclose(rows.stats, ref_stats, rtol=3e-6, atol=1e-3)
        gamma, beta = (1 + 0.2 * torch.randn(C, generator=g)).to(dev), (0.2 * torch.randn(C, generator=g)).to(dev)
        w = (torch.randn(N, C, generator=g) / C ** 0.5).to(dtype).to(dev)
        wf, cs, bf = ops.fold_layernorm(w, None, gamma, beta)
        y_f = ops.gemm(rows.x16, wf, bias=bf, ln_apply=(rows, cs, 1e-5))
        mu, var = h64.mean(1, keepdim=True), h64.var(1, unbiased=False, keepdim=True)
        z = ((h64 - mu) / (var + 1e-5).sqrt() * gamma.double() + beta.double()) @ w.double().T
        assert relerr(y_f, z.float()) < (1.2e-2 if dtype == torch.bfloat16 else 1.0e-3)
    finally:
        lib.sx_gemm_force_tile(-1)
    # a shape the cost model gives a lock-step tile: the call fails loudly (callers keep the separate sx_layernorm there)
    a = torch.randn(2048, 1280, generator=g).to(dtype).to(dev)
    w = torch.randn(1280, 1280, generator=g).to(dtype).to(dev)
    rows = ops.LnRows(2048, 1280, dtype, dev)
    with pytest.raises(RuntimeError, match="ping-pong"):
        ops.gemm(a, w, out_dtype=torch.float32, ln_emit=rows)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(1024, 1280, 1280), (128 * 300, 512, 704), (4096, 2560, 1344)])
def test_gemm_strip_kernel_matches_pingpong_producer(dev, dtype, M, N, K):
    """csrc/gemm_strip.hip (persistent 128-row strips, two accumulator sets, residual loads / stores under the main loop; opt-in:
    sx_gemm_force_tile(9) — it measured slower than the ping-pong producer, profiles/r6_ab_experiments.md §1): same LayerNorm-producer
    contract as gemm_pp.hip's. C and x16 bit for bit, row sums to fp32 summation noise and — written by plain stores in a fixed order,
    no atomics — the same bits in every launch. Shapes: 8 workgroups x 5 sub-tiles; 300 strips over 256 workgroups (some walk two),
    K / 32 = 22 phases (one rolled phase behind the 21 peeled ones); N = 2560 (the LDS bias table's limit)."""
    from seedx_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(41)
    a = (torch.randn(M, K, generator=g) * 0.5).to(dtype).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dtype).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    res = (torch.randn(M, N, generator=g) * 2.0 + torch.randn(M, 1, generator=g)).to(dev)
    try:
        assert lib.sx_gemm_force_tile(8) == 0
        r0 = ops.LnRows(M, N, dtype, dev)
        c0 = ops.gemm(a, w, bias=bias, residual=res, out_dtype=torch.float32, ln_emit=r0)
        assert lib.sx_gemm_force_tile(9) == 0
        outs = []
        for _ in range(3):
            r1 = ops.LnRows(M, N, dtype, dev)
            c1 = ops.gemm(a, w, bias=bias, residual=res, out_dtype=torch.float32, ln_emit=r1)
            outs.append((c1.clone(), r1.x16.clone(), r1.stats.clone()))
    finally:
        lib.sx_gemm_force_tile(-1)
    c1, x1, s1 = outs[0]
    assert torch.equal(c1, c0) and torch.equal(x1, r0.x16)
    h64 = c1.double()
    ref = torch.stack([h64.sum(1), (h64 * h64).sum(1)], dim=1)
    assert torch.allclose(s1, ref, rtol=3e-6, atol=1e-3), (s1 - ref).abs().max()
    for c, x, s in outs[1:]:
        assert torch.equal(c, c1) and torch.equal(x, x1) and torch.equal(s, s1)
