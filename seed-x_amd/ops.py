"""Thin torch-tensor wrappers over the C-ABI kernels (include/seedx_hip.h).

PyTorch is used only for device memory and the current HIP stream: every wrapper passes ``data_ptr()``s and
``torch.cuda.current_stream().cuda_stream`` to libseedx_hip.so. There is no eager/CPU fallback — a missing
library or a non-CUDA tensor raises.
"""
import ctypes as C
import math

import os

import torch

from . import _lib
from ._lib import (SX_A_CONV3X3, SX_A_LINEAR, SX_ACT_GELU, SX_ACT_NONE, SX_ACT_SILU, SX_BF16, SX_F16, SX_F32,
                   AttnArgs, AttnSmallArgs, GemmArgs, GemvArgs, check)

_DT = {torch.float16: SX_F16, torch.bfloat16: SX_BF16, torch.float32: SX_F32}
_TD = {v: k for k, v in _DT.items()}
ACT = {None: SX_ACT_NONE, "none": SX_ACT_NONE, "gelu": SX_ACT_GELU, "silu": SX_ACT_SILU}


def dt_code(dtype):
    return _DT[dtype]


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("seedx_amd.ops: tensors must live on the GPU (no CPU fallback)")
    return C.c_void_p(t.data_ptr())


def _f32c(t):
    """fp32 contiguous parameter tensor or None."""
    if t is None:
        return None
    assert t.dtype == torch.float32 and t.is_contiguous(), "expected contiguous fp32 tensor"
    return t


# ---------------------------------------------------------------------------------------------------------
# GEMM family
# ---------------------------------------------------------------------------------------------------------
class GnStats:
    """fp64 GroupNorm statistics [B, groups, 2] of a tensor that a GEMM / conv is about to produce: handed to the PRODUCER
    (``gemm(..., gn=...)`` / ``conv3x3(..., gn=...)``), whose epilogue accumulates them when it runs on a ping-pong tile (then
    ``ready`` is set), and to the CONSUMER ``groupnorm(..., stats=...)``, which then skips its own zero + statistics launches.
    ``buf`` must be zero before the producer runs (slices of one arena zeroed once per forward: ``GnStats.arena``)."""
    __slots__ = ("buf", "groups", "rows", "ready")

    def __init__(self, buf, groups, rows):
        self.buf, self.groups, self.rows, self.ready = buf, int(groups), int(rows), False

    @staticmethod
    def arena(n, B, groups, device):
        """n zeroed [B, groups, 2] fp64 slots with ONE fill launch."""
        return torch.zeros((n, B, groups, 2), dtype=torch.float64, device=device)


class LnRows:
    """A LayerNorm folded into the GEMMs either side of it (sx_gemm_ln): the PRODUCER (``gemm(..., ln_emit=rows)``: fp32 output)
    also writes ``x16`` = its output rounded to the operand dtype and adds the rows' (sum, sum of squares) to ``stats`` [M, 2]
    fp64 (zero beforehand); the CONSUMER (``gemm(rows.x16, w_folded, bias=b_folded, ln_apply=(rows, colsum, eps))``) applies
    (mu, rstd) per row in its epilogue. Both launches must run on ping-pong tiles: ``ln_fold_ok`` says so beforehand."""
    __slots__ = ("x16", "stats")

    def __init__(self, M, C, dtype, device, stats=None):
        self.x16 = torch.empty((M, C), dtype=dtype, device=device)
        self.stats = stats if stats is not None else torch.zeros((M, 2), dtype=torch.float64, device=device)
        assert self.stats.shape == (M, 2) and self.stats.dtype == torch.float64 and self.stats.is_contiguous()


def gemm_tile(M, N, K, glu=False, conv=False):
    """Tile config 0..8 the cost model gives an M x N x K problem (7 / 8 = the 256-row ping-pong tiles); host-only."""
    return int(_lib.load().sx_gemm_pick_tile(int(M), int(N), int(K), 1 if glu else 0, 1 if conv else 0))


def ln_fold_ok(M, C, consumers, producers):
    """True when every GEMM around a LayerNorm of width C over M rows runs on a ping-pong tile. consumers: [(N, glu)] of the
    projections behind the norm (K = C); producers: [K] of the fp32-output projections ahead of it (N = C)."""
    return all(gemm_tile(M, n, C, glu) >= 7 for n, glu in consumers) and all(gemm_tile(M, C, k) >= 7 for k in producers)


def fold_layernorm(w, bias, gamma, beta):
    """(w', colsum, bias') of a projection behind LayerNorm(gamma, beta): w' = w * gamma in w's dtype, colsum[n] = sum_k w'[n, k]
    (of the ROUNDED w': what the MFMA multiplies), bias' = bias + w beta. Per output row, so GLU-packed rows fold as they are."""
    wf = (w.float() * gamma.float()[None, :]).to(w.dtype).contiguous()
    cs = wf.float().sum(dim=1).contiguous()
    b = w.float() @ beta.float()
    if bias is not None:
        b = b + bias.float()
    return wf, cs, b.contiguous()


GN_FUSE = os.environ.get("SX_GN_FUSE", "1") != "0"     # A/B switch (tools/bench_unet_ab.py): 0 = every GroupNorm runs its own statistics pass


def _launch_gemm(lib, args, gn, what):
    if gn is None or not GN_FUSE:
        check(lib.sx_gemm(C.byref(args), _stream()), what)
        return
    assert gn.buf.dtype == torch.float64 and gn.buf.is_contiguous() and args.M % gn.rows == 0 \
        and gn.buf.numel() == (args.M // gn.rows) * gn.groups * 2, "GnStats does not describe this output"
    fused = C.c_int32(0)
    check(lib.sx_gemm_gn(C.byref(args), gn.buf.data_ptr(), gn.groups, gn.rows, C.byref(fused), _stream()), what)
    gn.ready = bool(fused.value)


def gemm(a, w, bias=None, bias2d=None, bias2d_rows=0, residual=None, res_mod=0, act=None, glu=False,
         out_dtype=None, out=None, n_valid=0, ld_bias2d=0, gn=None, ln_emit=None, ln_apply=None, a_planes=1):
    """out[M, N_out] = epilogue(a[M, K] @ w[N, K]^T). a, w: 16-bit contiguous. residual/bias fp32.
    gn: optional GnStats of the output (the next GroupNorm's statistics pass fused into this launch, see GnStats).
    ln_emit / ln_apply: the two sides of a folded LayerNorm (see LnRows).
    a_planes = 2: a is [M, 2K] = [hi | lo], the planes of an fp32-grade activation (split16 / rmsnorm_planes / attention_f32)."""
    lib = _lib.load()
    assert a.dim() == 2 and w.dim() == 2 and a.is_contiguous() and w.is_contiguous()
    assert a.dtype == w.dtype and a.dtype in (torch.float16, torch.bfloat16)
    M, K = a.shape
    N = w.shape[0]
    assert a_planes in (1, 2) and K % a_planes == 0
    K //= a_planes
    assert w.shape[1] == K, f"K mismatch {a.shape} vs {w.shape} (a_planes {a_planes})"
    n_out = N // 2 if glu else N
    n_store = n_valid if n_valid else n_out
    out_dtype = out_dtype or a.dtype
    if out is None:
        out = torch.empty((M, n_store), dtype=out_dtype, device=a.device)
    assert out.dtype == out_dtype and out.stride(-1) == 1 and out.shape[0] == M
    args = GemmArgs()
    args.A, args.W, args.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
    args.bias = _f32c(bias).data_ptr() if bias is not None else None
    if bias2d is not None:
        assert bias2d.dtype == torch.float32 and bias2d.stride(-1) == 1
        args.bias2d = bias2d.data_ptr()
        args.ld_bias2d = ld_bias2d or bias2d.stride(0)
    if residual is not None:
        assert residual.dtype == torch.float32 and residual.stride(-1) == 1
        args.residual = residual.data_ptr()
        args.ldr = residual.stride(0)
    args.M, args.N, args.K = M, N, K
    args.ldc = out.stride(0)
    args.n_valid = n_valid
    args.res_mod = res_mod
    args.bias2d_rows = bias2d_rows
    args.dtype = _DT[a.dtype]
    args.out_dtype = _DT[out_dtype]
    args.act = ACT[act]
    args.glu = 1 if glu else 0
    args.a_mode = SX_A_LINEAR
    args.a_planes = a_planes
    if ln_emit is not None or ln_apply is not None:
        assert gn is None and not (ln_emit is not None and ln_apply is not None)
        la = _lib.GemmLnArgs()
        if ln_emit is not None:
            assert ln_emit.x16.shape == (M, N) and ln_emit.x16.dtype == a.dtype and ln_emit.stats.shape[0] == M
            la.x16_out, la.ld_x16, la.row_stats_out = ln_emit.x16.data_ptr(), ln_emit.x16.stride(0), ln_emit.stats.data_ptr()
        else:
            rows, colsum, eps = ln_apply
            assert rows.stats.shape[0] == M and colsum.dtype == torch.float32 and colsum.numel() == N and colsum.is_contiguous()
            la.row_stats_in, la.colsum, la.dim, la.eps = rows.stats.data_ptr(), colsum.data_ptr(), K, float(eps)
        check(lib.sx_gemm_ln(C.byref(args), C.byref(la), _stream()), "sx_gemm_ln")
        return out
    _launch_gemm(lib, args, gn, "sx_gemm")
    return out


def conv3x3(x, w, bias=None, bias2d=None, residual=None, stride=1, upsample=False, out_dtype=None, act=None,
            n_valid=0, pad_mode=0, gn=None):
    """3x3 / pad 1 convolution as implicit GEMM. x: [B, H, W, Cin] 16-bit NHWC contiguous; w: [Cout, 9*Cin]
    ((ky,kx,cin)-ordered). Returns [B, Hout*Wout, Cout] (NHWC flattened). bias2d: [B, Cout] fp32 per-sample add.
    residual: fp32 [B*Hout*Wout, Cout]. pad_mode 1 = pad only bottom/right (VAE encoder's stride-2 convs)."""
    lib = _lib.load()
    assert x.dim() == 4 and x.is_contiguous() and w.is_contiguous()
    B, H, W, Cin = x.shape
    Cout, K = w.shape
    assert K == 9 * Cin
    hv, wv = (2 * H, 2 * W) if upsample else (H, W)
    padsum = 1 if pad_mode else 2
    Hout, Wout = (hv + padsum - 3) // stride + 1, (wv + padsum - 3) // stride + 1
    M = B * Hout * Wout
    out_dtype = out_dtype or x.dtype
    n_store = n_valid if n_valid else Cout
    out = torch.empty((M, n_store), dtype=out_dtype, device=x.device)
    args = GemmArgs()
    args.A, args.W, args.C = x.data_ptr(), w.data_ptr(), out.data_ptr()
    args.bias = _f32c(bias).data_ptr() if bias is not None else None
    args.n_valid = n_valid
    if bias2d is not None:
        assert bias2d.dtype == torch.float32 and bias2d.stride(-1) == 1 and bias2d.shape == (B, Cout)
        args.bias2d = bias2d.data_ptr()
        args.bias2d_rows = Hout * Wout
        args.ld_bias2d = bias2d.stride(0)
    if residual is not None:
        assert residual.dtype == torch.float32 and residual.stride(-1) == 1
        args.residual = residual.data_ptr()
        args.ldr = residual.stride(0)
    args.M, args.N, args.K = M, Cout, K
    args.ldc = n_store
    args.dtype = _DT[x.dtype]
    args.out_dtype = _DT[out_dtype]
    args.act = ACT[act]
    args.a_mode = SX_A_CONV3X3
    args.B, args.Hin, args.Win, args.Cin, args.Hout, args.Wout = B, H, W, Cin, Hout, Wout
    args.stride = stride
    args.upsample = 1 if upsample else 0
    args.pad_mode = pad_mode
    _launch_gemm(lib, args, gn, "sx_gemm(conv3x3)")
    return out.view(B, Hout * Wout, n_store)


def pack_decode_tiles(w):
    """Row-major [N, K] 16-bit weight → the decode layout [N/16][K/32][16][32] (same shape, other element order): every
    16-row x 32-k MFMA operand tile is 1 KB contiguous and the tiles of a row group follow each other along K, so the skinny
    GEMM's wave-wide loads are whole contiguous kilobytes instead of 64-B pieces of 16 rows 2K bytes apart."""
    N, K = w.shape
    assert N % 16 == 0 and K % 32 == 0
    return w.view(N // 16, 16, K // 32, 32).permute(0, 2, 1, 3).contiguous().view(N, K)


def pack_decode_tiles20(w):
    """Row-major [N, K] 16-bit weight → 20-row decode tiles [N/20][K/32][20][32] (sx_gemv w_layout 2): a 32-k slab of a 20-row
    group is 1280 contiguous bytes = the 1-KB MFMA operand tile of rows 0..15 followed by rows 16..19."""
    N, K = w.shape
    assert N % 20 == 0 and K % 32 == 0
    return w.view(N // 20, 20, K // 32, 32).permute(0, 2, 1, 3).contiguous().view(N, K)


class Tiled16:
    """A [rows <= 32, cols] 16-bit activation of the decode step held as MFMA operand tiles [rows/16][cols/32][16][32] (SX_TILED16
    in include/seedx_hip.h): what the skinny GEMM reads with one contiguous 1-KB load per operand. Rows 16..31 (lock-step batches
    above 16) are a second block of tiles behind the first; rows >= `rows` of the last block are padding."""
    __slots__ = ("t", "rows", "cols", "planes")

    def __init__(self, rows, cols, dtype, device, planes=1):
        """planes = 2: [2 planes][row blocks][cols/32][16][32] — the hi plane's row blocks, then the lo plane's, of an fp32-grade activation
        (sx_gemv x_planes = 2; 17..32 rows = two row blocks per plane, round 6)."""
        assert rows <= 32 and cols % 32 == 0 and planes in (1, 2)
        nb = planes * ((rows + 15) // 16)
        self.t, self.rows, self.cols, self.planes = torch.empty((nb, cols // 32, 16, 32), dtype=dtype, device=device), rows, cols, planes

    @property
    def dtype(self):
        return self.t.dtype

    def dense(self):
        """[rows, cols] row-major copy (tests); planes = 2: fp32 hi + lo."""
        nb = self.t.shape[0]
        d = self.t.permute(0, 2, 1, 3).reshape(nb * 16, self.cols)
        if self.planes == 2:
            half = (nb // 2) * 16
            return d[:self.rows].float() + d[half:half + self.rows].float()
        return d[:self.rows].contiguous()


def gemv(x, w, residual=None, act=None, glu=False, out_dtype=None, w_tiles=None, y_tiled=False, workspace=None,
         emit_norm=False, ssq_in=None, w_tiles20=None, planes_out=False, norm_gamma=None):
    """w_tiles: the same weight in the decode layout (pack_decode_tiles); used instead of w when the MFMA path runs.
    x may be a Tiled16 (then w_tiles is required); y_tiled returns the 16-bit result as a Tiled16 for the next gemv.
    workspace: zero-initialised uint8 scratch enabling split-K over workgroups for shapes that need it (sx_gemv_args.workspace).
    RMSNorm fold (sx_gemv_args.x16_out / row_ssq_*; MFMA path, tiled x): ``emit_norm`` (fp32 residual outputs) → returns
    (y, x16, ssq): y also as 16-bit operand tiles and the rows' sums of squares per workgroup; ``ssq_in`` = (ssq, dim, eps) of
    the producer → the accumulators are scaled by rsqrt(sum(ssq) / dim + eps) (gamma lives in this launch's weights).
    w_tiles20: the weight as 20-row decode tiles (pack_decode_tiles20) instead of w_tiles: one workgroup per 20 rows.
    planes_out (M <= 16): the tiled 16-bit result (y_tiled) or the emit_norm x16 is written as two planes (Tiled16 planes = 2) for a next
    gemv with fp32-grade x; norm_gamma (with emit_norm): x16 = planes of y * gamma — the NEXT RMSNorm's weight on the activation side, so
    the projection behind the norm keeps exact weights and only applies rstd (ssq_in)."""
    lib = _lib.load()
    xt = isinstance(x, Tiled16)
    if xt:
        M, K = x.rows, x.cols
        assert x.dtype == w.dtype and (x.planes == 2 or w_tiles is not None or w_tiles20 is not None)
    else:
        assert x.dim() == 2 and x.is_contiguous() and w.is_contiguous() and x.dtype == w.dtype
        M, K = x.shape
    N = w.shape[0]
    n_out = N // 2 if glu else N
    out_dtype = out_dtype or x.dtype
    dev = x.t.device if xt else x.device
    if y_tiled:
        yt = Tiled16(M, n_out, out_dtype, dev, planes=2 if planes_out else 1)
        y = yt.t
    else:
        y = torch.empty((M, n_out), dtype=out_dtype, device=dev)
    args = GemvArgs()
    args.x, args.W, args.y = (x.t if xt else x).data_ptr(), w.data_ptr(), y.data_ptr()
    args.x_layout = 1 if xt else 0
    args.x_planes = x.planes if xt else 1
    if workspace is not None:
        args.workspace, args.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    if residual is not None:
        assert residual.dtype == torch.float32 and residual.is_contiguous() and residual.shape == (M, n_out)
        args.residual = residual.data_ptr()
    args.M, args.N, args.K = M, N, K
    args.dtype, args.out_dtype, args.act, args.glu = _DT[x.dtype], _DT[out_dtype], ACT[act], 1 if glu else 0
    if y_tiled:
        args.out_dtype |= _lib.SX_TILED16
    if planes_out:
        assert y_tiled or emit_norm
        args.out_planes = 1
    if w_tiles is not None and (xt or y_tiled or M >= 5) and K % 64 == 0 and K >= 256 and N % 32 == 0:
        assert w_tiles.shape == w.shape and w_tiles.dtype == w.dtype and w_tiles.is_contiguous()
        args.W, args.w_layout = w_tiles.data_ptr(), 1
    if w_tiles20 is not None and not glu and (xt or y_tiled or M >= 5) and K % 64 == 0 and K >= 256 and N % 32 == 0 and N % 20 == 0:
        assert w_tiles20.shape == w.shape and w_tiles20.dtype == w.dtype and w_tiles20.is_contiguous()
        args.W, args.w_layout = w_tiles20.data_ptr(), 2
    x16 = ssq = None
    if emit_norm:
        assert args.w_layout in (1, 2) and out_dtype == torch.float32 and not glu and not y_tiled
        x16 = Tiled16(M, n_out, w.dtype, dev, planes=2 if planes_out else 1)
        if norm_gamma is not None:
            assert norm_gamma.dtype == torch.float32 and norm_gamma.is_contiguous() and norm_gamma.numel() == n_out
            args.x16_gamma = norm_gamma.data_ptr()
        ssq = torch.empty((16 * ((M + 15) // 16), lib.sx_gemv_ssq_parts(N, 0, args.w_layout)), dtype=torch.float32, device=dev)   # [row][workgroup]
        args.x16_out, args.row_ssq_out = x16.t.data_ptr(), ssq.data_ptr()
    if ssq_in is not None:
        t, dim, eps = ssq_in
        assert args.w_layout == 1 and t.dtype == torch.float32 and t.is_contiguous() and t.shape[0] == 16 * ((M + 15) // 16)
        args.row_ssq_in, args.ssq_in_parts, args.ssq_dim, args.ssq_eps = t.data_ptr(), t.shape[1], int(dim), float(eps)
    check(lib.sx_gemv(C.byref(args), _stream()), "sx_gemv")
    out = yt if y_tiled else y
    return (out, x16, ssq) if emit_norm else out


def linear(x, w, **kw):
    """Dispatch M <= 16 rows to the weight-streaming GEMV / skinny GEMM (when the epilogue allows), else the MFMA GEMM."""
    M, K, N = x.shape[0], x.shape[1], w.shape[0]
    # 9..16 rows only exist on the MFMA skinny kernel (K % 64 == 0, K >= 256, N % 32 == 0); other shapes keep the tiled GEMM
    skinny_ok = M <= 8 or (K % 64 == 0 and K >= 256 and N % 32 == 0)
    if M <= 16 and skinny_ok and kw.get("bias") is None and kw.get("bias2d") is None and not kw.get("res_mod") \
            and kw.get("out") is None and not kw.get("n_valid"):
        res = kw.get("residual")
        if res is None or res.is_contiguous():
            return gemv(x, w, residual=res, act=kw.get("act"), glu=kw.get("glu", False), out_dtype=kw.get("out_dtype"),
                        w_tiles=kw.get("w_tiles"))
    kw.pop("w_tiles", None)
    return gemm(x, w, **kw)


# ---------------------------------------------------------------------------------------------------------
# Norms
# ---------------------------------------------------------------------------------------------------------
def layernorm(x, gamma, beta, eps, out_dtype, rms=False, tiled=False):
    """tiled (rows <= 32, 16-bit out): the result as a Tiled16 — the decode step's norms feed skinny GEMMs only."""
    lib = _lib.load()
    assert x.is_contiguous()
    cols = x.shape[-1]
    rows = x.numel() // cols
    if tiled:
        yt = Tiled16(rows, cols, out_dtype, x.device)
        y, code = yt.t, _DT[out_dtype] | _lib.SX_TILED16
    else:
        y, code = torch.empty(x.shape, dtype=out_dtype, device=x.device), _DT[out_dtype]
    check(lib.sx_layernorm(_p(x), _DT[x.dtype], _p(y), code, _p(_f32c(gamma)), _p(_f32c(beta)), rows, cols,
                           float(eps), 1 if rms else 0, _stream()), "sx_layernorm")
    return yt if tiled else y


def rmsnorm(x, gamma, eps, out_dtype, tiled=False):
    return layernorm(x, gamma, None, eps, out_dtype, rms=True, tiled=tiled)


def groupnorm(x, gamma, beta, groups, eps, silu, out_dtype, want_raw=False, x2=None, comm=None, hw_total=None, planes=False,
              stats=None):
    """x: fp32 [B, HW, C] (NHWC). Returns y (16-bit) and optionally a 16-bit raw copy of x.
    stats: GnStats filled by x's producer (``ready``): only the apply pass runs (single rank, no x2).
    planes: True / 3: y (and raw) come out as bf16 [B, HW, 3C] rows [hi | hi | lo] (split_bf16 role "a") instead; 2: as fp16
    [B, HW, 2C] rows [hi | lo] (split16's layout: the operand of a conv whose fp16 weight rows are duplicated per tap).
    x2: optional second fp32 [B, HW, C2] tensor — the op then runs over the channel concatenation [x | x2] (never built).
    comm / hw_total: x holds only this rank's pixel rows of images with hw_total rows (seqpar.py): the fp64 statistics are
    all-reduced over the ranks between the statistics pass and the apply pass."""
    lib = _lib.load()
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 3
    B, HW, C1 = x.shape
    Cc = C1
    if x2 is not None:
        assert x2.dtype == torch.float32 and x2.is_contiguous() and x2.shape[:2] == (B, HW)
        Cc = C1 + x2.shape[2]
    if planes == 2:
        out_dtype, code, cols = torch.float16, _lib.SX_F16X2, 2 * Cc
    elif planes:
        out_dtype, code, cols = torch.bfloat16, _lib.SX_BF16X3, 3 * Cc
    else:
        code, cols = _DT[out_dtype], Cc
    y = torch.empty((B, HW, cols), dtype=out_dtype, device=x.device)
    raw = torch.empty((B, HW, cols), dtype=out_dtype, device=x.device) if want_raw else None
    if stats is not None and stats.ready and x2 is None and (comm is None or comm.world == 1):
        assert stats.groups == groups and stats.rows == HW and stats.buf.numel() == B * groups * 2
        check(lib.sx_groupnorm_sp(_p(x), None, C1, _p(y), _p(raw), code, _p(_f32c(gamma)), _p(_f32c(beta)), _p(stats.buf), B, HW, HW,
                                  Cc, groups, float(eps), 1 if silu else 0, 2, _stream()), "sx_groupnorm_sp(apply, fused statistics)")
        return (y, raw) if want_raw else y
    stats = torch.empty((B, groups, 2), dtype=torch.float64, device=x.device)
    if comm is None or comm.world == 1:
        check(lib.sx_groupnorm2(_p(x), _p(x2), C1, _p(y), _p(raw), code, _p(_f32c(gamma)), _p(_f32c(beta)), _p(stats),
                                B, HW, Cc, groups, float(eps), 1 if silu else 0, _stream()), "sx_groupnorm")
    else:
        args = (_p(x), _p(x2), C1, _p(y), _p(raw), code, _p(_f32c(gamma)), _p(_f32c(beta)), _p(stats), B, HW,
                int(hw_total), Cc, groups, float(eps), 1 if silu else 0)
        check(lib.sx_groupnorm_sp(*args, 1, _stream()), "sx_groupnorm_sp(stats)")
        comm.all_reduce(stats)
        check(lib.sx_groupnorm_sp(*args, 2, _stream()), "sx_groupnorm_sp(apply)")
    return (y, raw) if want_raw else y


def softmax_rows(x, scale, out_dtype):
    """softmax(scale · x) over the last dim; x fp32 [R, C] (row stride free) → 16-bit (or fp32) [R, C]."""
    lib = _lib.load()
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    R, Cc = x.shape
    y = torch.empty((R, Cc), dtype=out_dtype, device=x.device)
    check(lib.sx_softmax_rows(_p(x), x.stride(0), _p(y), y.stride(0), R, Cc, float(scale), _DT[out_dtype], _stream()),
          "sx_softmax_rows")
    return y


# ---------------------------------------------------------------------------------------------------------
# Attention
# ---------------------------------------------------------------------------------------------------------
def _bshd_strides(t):
    assert t.dim() == 4 and t.stride(3) == 1, "expected a [B, S, H, D] view with unit stride on D"
    return t.stride(0), t.stride(1), t.stride(2)


def attention(q, k, v, scale, causal=False, out=None):
    """Flash attention on MFMA. q: [B, Sq, H, D] view, k/v: [B, Skv, H, D] views (any strides that are multiples of 8
    elements, unit D stride). V is read in this natural layout (the kernel transposes fragments with ds_read_b64_tr_b16).
    Returns [B, Sq, H*D] 16-bit."""
    lib = _lib.load()
    B, Sq, H, D = q.shape
    Skv = k.shape[1]
    assert k.shape == (B, Skv, H, D) and v.shape == (B, Skv, H, D) and q.dtype == k.dtype == v.dtype
    if out is None:
        out = torch.empty((B, Sq, H * D), dtype=q.dtype, device=q.device)
    a = AttnArgs()
    a.Q, a.K, a.V, a.O = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
    a.B, a.H, a.Sq, a.Skv, a.D = B, H, Sq, Skv, D
    a.q_batch_stride, a.q_row_stride, a.q_head_stride = _bshd_strides(q)
    a.k_batch_stride, a.k_row_stride, a.k_head_stride = _bshd_strides(k)
    a.v_batch_stride, a.v_row_stride, a.v_head_stride = _bshd_strides(v)
    a.o_batch_stride, a.o_row_stride = out.stride(0), out.stride(1)
    a.scale = float(scale)
    a.causal = 1 if causal else 0
    a.dtype = _DT[q.dtype]
    check(lib.sx_attention(C.byref(a), _stream()), "sx_attention")
    return out


def attention_small(q, k, v, scale):
    lib = _lib.load()
    B, Sq, H, D = q.shape
    Skv = k.shape[1]
    out = torch.empty((B, Sq, H * D), dtype=q.dtype, device=q.device)
    a = AttnSmallArgs()
    a.Q, a.K, a.V, a.O = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
    a.B, a.H, a.Sq, a.Skv, a.D = B, H, Sq, Skv, D
    a.q_batch_stride, a.q_row_stride, a.q_head_stride = _bshd_strides(q)
    a.k_batch_stride, a.k_row_stride, a.k_head_stride = _bshd_strides(k)
    a.v_batch_stride, a.v_row_stride, a.v_head_stride = _bshd_strides(v)
    a.o_batch_stride, a.o_row_stride = out.stride(0), out.stride(1)
    a.scale = float(scale)
    a.dtype = _DT[q.dtype]
    check(lib.sx_attention_small(C.byref(a), _stream()), "sx_attention_small")
    return out


def attn_decode(q, kcache, vcache, ctx_len_dev, scale, nsplit=8):
    """q: [H, D] 16-bit; caches [H, Tmax, D]; ctx_len_dev: int32 device scalar. Returns [1, H*D]."""
    lib = _lib.load()
    H, D = q.shape
    Tmax = kcache.shape[1]
    out = torch.empty((1, H * D), dtype=q.dtype, device=q.device)
    scratch = torch.empty((H, nsplit, D + 2), dtype=torch.float32, device=q.device)
    check(lib.sx_attn_decode(_p(q), _p(kcache), _p(vcache), _p(out), _p(scratch), _p(ctx_len_dev), H, D, Tmax, nsplit,
                             float(scale), _DT[q.dtype], _stream()), "sx_attn_decode")
    return out


# ---------------------------------------------------------------------------------------------------------
# fp32-grade activations of the Llama decoder (LlamaForCausalLM(precise=True); csrc/precise.hip)
# ---------------------------------------------------------------------------------------------------------
def _planes_out(rows, cols, dtype, device, tiled):
    if tiled:
        t = Tiled16(rows, cols, dtype, device, planes=2)
        return t, t.t, _DT[dtype] | _lib.SX_TILED16
    y = torch.empty((rows, 2 * cols), dtype=dtype, device=device)
    return y, y, _DT[dtype]


def split16(x, dtype, tiled=False):
    """fp32 [rows, cols] → its two 16-bit planes x = hi + lo: [rows, 2*cols] = [hi | lo] (gemm a_planes = 2), or with ``tiled`` a
    two-block Tiled16 (gemv)."""
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    rows, cols = x.shape
    ret, buf, code = _planes_out(rows, cols, dtype, x.device, tiled)
    check(_lib.load().sx_split16(_p(x), x.stride(0), _p(buf), rows, cols, code, _stream()), "sx_split16")
    return ret


def rmsnorm_planes(x, gamma, eps, dtype, tiled=False, want_f32=False, want_planes=True):
    """LlamaRMSNorm in fp32 → (planes of y or None, y fp32 or None)."""
    assert x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous()
    rows, cols = x.shape
    ret = buf = None
    code = _DT[dtype]
    if want_planes:
        ret, buf, code = _planes_out(rows, cols, dtype, x.device, tiled)
    y32 = torch.empty_like(x) if want_f32 else None
    check(_lib.load().sx_rmsnorm_planes(_p(x), _p(_f32c(gamma)), _p(y32), _p(buf), rows, cols, float(eps), code, _stream()),
          "sx_rmsnorm_planes")
    return ret, y32


def rope_kv_append_f32(qkv, kcache, vcache, cos_tab, sin_tab, pos_dev, G, T, H, D, table_dtype):
    """qkv fp32 [G*T, 3HD] (q rotated in place); caches [G, H, Tmax, D]: k fp32, v fp32 or — the mixed cache — the model's 16-bit dtype
    (= table_dtype); pos_dev int32 [G]."""
    assert qkv.dtype == torch.float32 and qkv.is_contiguous() and qkv.shape == (G * T, 3 * H * D)
    assert kcache.dtype == torch.float32 and vcache.dtype in (torch.float32, table_dtype) and kcache.dim() == 4 and kcache.shape[0] == G \
        and kcache[0].is_contiguous() and vcache.stride() == kcache.stride()
    fn = "sx_rope_kv_append_f32" if vcache.dtype == torch.float32 else "sx_rope_kv_append_f32_v16"
    check(getattr(_lib.load(), fn)(_p(qkv), _p(kcache), _p(vcache), _p(cos_tab), _p(sin_tab), _p(pos_dev), G, T, H, D,
                                   kcache.shape[2], kcache.stride(0), _DT[table_dtype], _stream()), fn)


def attention_f32(qkv, kcache, vcache, pos_dev, G, T, H, D, scale, dtype, tiled=False, nsplit=1, scratch=None, rope=None):
    """Causal fp32 attention of a T-token chunk per sequence over the fp32 cache (row t sees keys 0 .. pos[g] + t); q = the rotated
    head rows at the front of qkv's rows. Returns the planes of the context [G*T, H*D] (see split16).
    rope=(cos, sin) (T == 1, D == 128): qkv holds the UNROTATED q | k | v rows of the new token; the launch rotates q and k, appends k / v
    to the caches at pos[g] and attends — rope_kv_append_f32 + attention_f32 in one launch per layer."""
    assert qkv.dtype == torch.float32 and qkv.is_contiguous() and qkv.shape[0] == G * T and qkv.shape[1] >= H * D
    assert kcache.dtype == torch.float32 and kcache.shape[0] == G and kcache.shape[1] == H and kcache.shape[3] == D
    assert vcache.dtype in (torch.float32, dtype) and vcache.stride() == kcache.stride()
    ret, buf, code = _planes_out(G * T, H * D, dtype, qkv.device, tiled)
    a = _lib.AttnF32Args()
    a.q, a.kcache, a.vcache, a.out, a.pos0_dev = _p(qkv), _p(kcache), _p(vcache), _p(buf), _p(pos_dev)
    a.q_row_stride, a.cache_seq_stride = qkv.stride(0), kcache.stride(0)
    a.G, a.T, a.H, a.D, a.Tmax, a.dtype, a.scale, a.causal = G, T, H, D, kcache.shape[2], code, float(scale), 1
    a.v16 = 0 if vcache.dtype == torch.float32 else 1
    if nsplit > 1 and T == 1:         # decode step of few sequences: key splits (scratch: fp32 [G, H, nsplit, D + 2])
        if scratch is None:
            scratch = torch.empty((G, H, nsplit, D + 2), dtype=torch.float32, device=qkv.device)
        assert scratch.dtype == torch.float32 and scratch.numel() >= G * H * nsplit * (D + 2)
        a.nsplit, a.scratch = nsplit, _p(scratch)
    if rope is not None:
        assert T == 1 and D == 128 and qkv.shape[1] == 3 * H * D
        cos, sin = rope
        a.rope_cos, a.rope_sin = _p(cos), _p(sin)
        a.k_new, a.v_new = qkv.data_ptr() + 4 * H * D, qkv.data_ptr() + 8 * H * D
    check(_lib.load().sx_attention_f32(C.byref(a), _stream()), "sx_attention_f32")
    return ret


def attention_f32_full(q, k, v, scale, dtype):
    """Non-causal fp32 attention (fp32 FMA arithmetic and softmax): q [B, Sq, H, D], k / v [B, Skv, H, D] fp32 views with unit D stride
    (e.g. slices of one fused projection output). Returns the planes of the context [B*Sq, H*D] (split16's row layout)."""
    B, Sq, H, D = q.shape
    Skv = k.shape[1]
    assert q.dtype == k.dtype == v.dtype == torch.float32 and k.shape == (B, Skv, H, D) and v.shape == k.shape
    assert q.stride(3) == 1 and q.stride(2) == D and q.stride(0) == Sq * q.stride(1), "q rows must be [B*Sq] rows of H*D heads"
    assert k.stride(3) == 1 and v.stride() == k.stride()
    ret, buf, code = _planes_out(B * Sq, H * D, dtype, q.device, False)
    a = _lib.AttnF32Args()
    a.q, a.kcache, a.vcache, a.out, a.pos0_dev = _p(q), _p(k), _p(v), _p(buf), None
    a.q_row_stride, a.cache_seq_stride, a.kv_row_stride, a.kv_head_stride = q.stride(1), k.stride(0), k.stride(1), k.stride(2)
    a.G, a.T, a.H, a.D, a.Tmax, a.dtype, a.scale, a.causal = B, Sq, H, D, Skv, code, float(scale), 0
    check(_lib.load().sx_attention_f32(C.byref(a), _stream()), "sx_attention_f32")
    return ret


def linear_planes(x32, w, w_tiles=None, **kw):
    """epilogue(x32 @ w^T) with x32 fp32 carried as two 16-bit planes: <= 16 rows on the weight-streaming skinny GEMM, else the MFMA GEMM."""
    M, K = x32.shape
    N = w.shape[0]
    if M <= 32 and K % 64 == 0 and K >= 256 and N % 32 == 0:
        return gemv(split16(x32, w.dtype, tiled=True), w, w_tiles=w_tiles, **kw)
    return gemm(split16(x32, w.dtype), w, a_planes=2, **kw)


# ---------------------------------------------------------------------------------------------------------
# LLM glue
# ---------------------------------------------------------------------------------------------------------
def rope_kv_append(qkv, kcache, vcache, cos_tab, sin_tab, pos0_dev, H, D):
    lib = _lib.load()
    T = qkv.shape[0]
    assert qkv.is_contiguous() and qkv.shape[1] == 3 * H * D
    check(lib.sx_rope_kv_append(_p(qkv), _p(kcache), _p(vcache), _p(cos_tab), _p(sin_tab), _p(pos0_dev), T, H, D,
                                kcache.shape[1], _DT[qkv.dtype], _stream()), "sx_rope_kv_append")


def embedding(ids_i32, table):
    lib = _lib.load()
    T = ids_i32.numel()
    out = torch.empty((T, table.shape[1]), dtype=torch.float32, device=table.device)
    check(lib.sx_embedding(_p(ids_i32), _p(table), _p(out), T, table.shape[1], _DT[table.dtype], _stream()),
          "sx_embedding")
    return out


def scatter_rows(src, rows_i32, dst):
    lib = _lib.load()
    assert src.dtype == torch.float32 and dst.dtype == torch.float32 and src.is_contiguous() and dst.is_contiguous()
    check(lib.sx_scatter_rows(_p(src), _p(rows_i32), _p(dst), src.shape[0], src.shape[1], _stream()),
          "sx_scatter_rows")


def greedy_next(logits, vocab, img_ids_dev, prev_id_dev, next_id_dev, out_ids=None, step_dev=None):
    lib = _lib.load()
    assert logits.dtype == torch.float32
    cap = out_ids.numel() if out_ids is not None else 0          # capacity: the kernel drops writes at step >= cap
    check(lib.sx_greedy_next_b(_p(logits), 0, vocab, _p(img_ids_dev), img_ids_dev.numel(), _p(prev_id_dev),
                               _p(next_id_dev), _p(out_ids), cap, _p(step_dev), 1, _stream()), "sx_greedy_next")


# ---------------------------------------------------------------------------------------------------------
# Elementwise / layout
# ---------------------------------------------------------------------------------------------------------
def cast(x, dtype):
    lib = _lib.load()
    assert x.is_contiguous()
    y = torch.empty(x.shape, dtype=dtype, device=x.device)
    check(lib.sx_cast(_p(x), _DT[x.dtype], _p(y), _DT[dtype], x.numel(), _stream()), "sx_cast")
    return y


# Accounting hint, not behaviour: while > 1, the K of the sx_gemm launches being issued is this many times the algorithmic K
# (the VAE's fp32-grade mode triples K to carry two bf16 planes per operand). bench.py's roofline pass divides by it so that
# "achieved" counts the FLOPs of the fp32 product being emulated, not the MFMA work spent on it.
OPERAND_PLANES = 1


def split_bf16(x, role="a"):
    """fp32 [..., C] → bf16 [..., 3C]: the planes of x = hi + lo laid out [hi | hi | lo] (role "a": rows of a GEMM's A
    operand) or [hi | lo | hi] (role "w": rows of its W operand). One GEMM over K = 3C then yields Ah·Wh + Ah·Wl + Al·Wh
    with fp32 accumulation — 16 mantissa bits per operand (the VAE's fp32-grade mode)."""
    lib = _lib.load()
    assert x.dtype == torch.float32 and x.is_contiguous() and x.shape[-1] % 4 == 0
    Cc = x.shape[-1]
    out = torch.empty(x.shape[:-1] + (3 * Cc,), dtype=torch.bfloat16, device=x.device)
    check(lib.sx_split_bf16(_p(x), _p(out), x.numel() // Cc, Cc, {"a": 0, "w": 1}[role], _stream()), "sx_split_bf16")
    return out


def copy2d(src, dst, dst_col_off=0):
    """dst[:, off:off+cols] = src (fp32 2-D, row strides honoured)."""
    lib = _lib.load()
    assert src.dtype == torch.float32 and dst.dtype == torch.float32 and src.dim() == 2 and dst.dim() == 2
    assert src.stride(1) == 1 and dst.stride(1) == 1
    d = dst[:, dst_col_off:dst_col_off + src.shape[1]]
    check(lib.sx_copy2d_f32(_p(src), src.stride(0), C.c_void_p(d.data_ptr()), dst.stride(0), src.shape[0],
                            src.shape[1], _stream()), "sx_copy2d_f32")


def add(a, b):
    lib = _lib.load()
    assert a.dtype == torch.float32 and b.dtype == torch.float32 and a.is_contiguous() and b.is_contiguous()
    y = torch.empty_like(a)
    check(lib.sx_add_f32(_p(a), _p(b), _p(y), a.numel(), _stream()), "sx_add_f32")
    return y


def patchify(img, patch, kpad, dtype):
    lib = _lib.load()
    assert img.dtype == torch.float32 and img.is_contiguous() and img.shape[1] == 3 and img.shape[2] == img.shape[3]
    B, _, S, _ = img.shape
    g = S // patch
    out = torch.empty((B * g * g, kpad), dtype=dtype, device=img.device)
    check(lib.sx_patchify(_p(img), _p(out), B, S, patch, kpad, _DT[dtype], _stream()), "sx_patchify")
    return out


def im2col3x3_small(x, kpad, dtype):
    lib = _lib.load()
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4
    B, H, W, Cin = x.shape
    out = torch.empty((B * H * W, kpad), dtype=dtype, device=x.device)
    check(lib.sx_im2col3x3_small(_p(x), _p(out), B, H, W, Cin, kpad, _DT[dtype], _stream()), "sx_im2col3x3_small")
    return out


def avgpool_tokens(x, k):
    lib = _lib.load()
    assert x.dtype == torch.float32 and x.is_contiguous()
    B, L, D = x.shape
    y = torch.empty((B, L // k, D), dtype=torch.float32, device=x.device)
    check(lib.sx_avgpool_tokens(_p(x), _p(y), B, L, D, k, _stream()), "sx_avgpool_tokens")
    return y


def timestep_embedding(t, dim, dtype, idx_dev=None, n=None):
    lib = _lib.load()
    assert t.dtype == torch.float32
    n = n if n is not None else t.numel()
    out = torch.empty((n, dim), dtype=dtype, device=t.device)
    check(lib.sx_timestep_embedding(_p(t), _p(idx_dev), _p(out), n, dim, _DT[dtype], _stream()),
          "sx_timestep_embedding")
    return out


def nchw_to_nhwc(src, ld=None, dst=None):
    lib = _lib.load()
    B, Cc, H, W = src.shape
    ld = ld or Cc
    if dst is None:
        dst = torch.zeros((B, H * W, ld), dtype=torch.float32, device=src.device)
    check(lib.sx_nchw_to_nhwc(_p(src.contiguous()), _p(dst), ld, B, Cc, H * W, _stream()), "sx_nchw_to_nhwc")
    return dst


def nhwc_to_nchw(src, Cc, H, W):
    lib = _lib.load()
    B = src.shape[0]
    ld = src.shape[-1]
    dst = torch.empty((B, Cc, H, W), dtype=torch.float32, device=src.device)
    check(lib.sx_nhwc_to_nchw(_p(src), ld, _p(dst), B, Cc, H * W, _stream()), "sx_nhwc_to_nchw")
    return dst


def silu_cast(x, dtype):
    lib = _lib.load()
    assert x.dtype == torch.float32 and x.is_contiguous()
    y = torch.empty(x.shape, dtype=dtype, device=x.device)
    check(lib.sx_silu_cast(_p(x), _p(y), _DT[dtype], x.numel(), _stream()), "sx_silu_cast")
    return y


def add_i32(p, delta):
    check(_lib.load().sx_add_i32_n(_p(p), int(delta), p.numel(), _stream()), "sx_add_i32_n")


# ---- lock-step batched decode (G sequences, one token each) ---------------------------------------------------------
def rope_kv_append_b(qkv, kcache, vcache, cos_tab, sin_tab, pos_dev, G, T, H, D):
    """qkv [G*T, 3HD]; kcache/vcache [G, H, Tmax, D]; pos_dev int32 [G]."""
    lib = _lib.load()
    assert qkv.is_contiguous() and qkv.shape == (G * T, 3 * H * D) and kcache.is_contiguous() and kcache.shape[0] == G
    check(lib.sx_rope_kv_append_b(_p(qkv), _p(kcache), _p(vcache), _p(cos_tab), _p(sin_tab), _p(pos_dev), G, T, H, D,
                                  kcache.shape[2], kcache.stride(0), _DT[qkv.dtype], _stream()), "sx_rope_kv_append_b")


def attn_decode_b(q, kcache, vcache, ctx_dev, scale, nsplit=8, out_tiled=False):
    """q [G, H, D] (rows may be strided views into the fused qkv projection); caches [G, H, Tmax, D]; ctx_dev int32 [G].
    Returns [G, H*D] (out_tiled: as a Tiled16 for the o-projection)."""
    lib = _lib.load()
    G, H, D = q.shape
    assert q.stride(2) == 1 and q.stride(1) == D and q.stride(0) % 8 == 0
    code = _DT[q.dtype]
    if out_tiled:
        ot = Tiled16(G, H * D, q.dtype, q.device)
        out, code = ot.t, code | _lib.SX_TILED16
    else:
        out = torch.empty((G, H * D), dtype=q.dtype, device=q.device)
    scratch = torch.empty((G, H, nsplit, D + 2), dtype=torch.float32, device=q.device)
    check(lib.sx_attn_decode_b(_p(q), _p(kcache), _p(vcache), _p(out), _p(scratch), _p(ctx_dev), G, H, D, kcache.shape[2],
                               kcache.stride(0), nsplit, float(scale), code, q.stride(0), _stream()), "sx_attn_decode_b")
    return ot if out_tiled else out


def attn_decode_fused(qkv, kcache, vcache, pos_dev, cos_tab, sin_tab, scale, H, D, counters, nsplit=8, out_tiled=False):
    """RoPE + KV append + split-KV attention + combine for ONE new token of each of G sequences, one launch.
    qkv [G, 3*H*D] (left untouched); caches [G, H, Tmax, D]; pos_dev int32 [G]; counters: zero int32 [>= G*H] (left zero).
    Returns [G, H*D] (out_tiled: as a Tiled16 for the o-projection)."""
    lib = _lib.load()
    G = qkv.shape[0]
    assert qkv.is_contiguous() and qkv.shape[1] == 3 * H * D and counters.dtype == torch.int32 and counters.numel() >= G * H
    code = _DT[qkv.dtype]
    if out_tiled:
        ot = Tiled16(G, H * D, qkv.dtype, qkv.device)
        out, code = ot.t, code | _lib.SX_TILED16
    else:
        out = torch.empty((G, H * D), dtype=qkv.dtype, device=qkv.device)
    scratch = torch.empty((G, H, nsplit, D + 2), dtype=torch.float32, device=qkv.device)
    a = _lib.AttnDecodeArgs()
    a.qkv, a.kcache, a.vcache, a.out, a.scratch, a.counters = _p(qkv), _p(kcache), _p(vcache), _p(out), _p(scratch), _p(counters)
    a.cos_tab, a.sin_tab, a.pos_dev = _p(cos_tab), _p(sin_tab), _p(pos_dev)
    a.G, a.H, a.D, a.Tmax, a.nsplit, a.dtype = G, H, D, kcache.shape[2], nsplit, code
    a.cache_seq_stride, a.scale = kcache.stride(0), float(scale)
    check(lib.sx_attn_decode_fused(C.byref(a), _stream()), "sx_attn_decode_fused")
    return ot if out_tiled else out


def greedy_next_b(logits, vocab, img_ids_dev, cur_dev, out_ids, step_dev):
    """logits fp32 [G, ld]; cur_dev int32 [G] (in: previous id, out: next id); out_ids int32 [G, ld_out]; step_dev [G]."""
    lib = _lib.load()
    G = logits.shape[0]
    assert logits.dtype == torch.float32 and logits.stride(1) == 1
    check(lib.sx_greedy_next_b(_p(logits), logits.stride(0), vocab, _p(img_ids_dev), img_ids_dev.numel(), _p(cur_dev),
                               _p(cur_dev), _p(out_ids), out_ids.stride(0) if out_ids is not None else 0, _p(step_dev), G,
                               _stream()), "sx_greedy_next_b")


def scatter_rows_step(src, step_dev, dst):
    """dst [G, rows, dim] fp32; dst[g, step[g]] = src[g]."""
    lib = _lib.load()
    G, rows, dim = dst.shape
    assert src.shape == (G, dim) and src.is_contiguous() and dst.is_contiguous()
    check(lib.sx_scatter_rows_step(_p(src), _p(step_dev), _p(dst), G, dim, rows, _stream()), "sx_scatter_rows_step")


def cfg_euler_step(eps, latents, scaled_next, sigmas_dev, step_dev, nb, C_lat, ld_scaled, gs, igs, mode):
    lib = _lib.load()
    n = latents.numel()
    check(lib.sx_cfg_euler_step(_p(eps), _p(latents), _p(scaled_next), _p(sigmas_dev), _p(step_dev), nb, n, C_lat,
                                ld_scaled, float(gs), float(igs), mode, _stream()), "sx_cfg_euler_step")
