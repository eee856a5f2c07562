"""Pixel-row ("sequence") sharding of ONE SDXL-UNet forward over the GPUs of a node (latency mode).

Why rows and not channels: xGMI is point-to-point (7 links x ~153 GB/s per GPU), so the sharding is chosen by the bytes it
moves. Megatron-style channel sharding of this UNet needs ≈235 all-reduces of the fp32 activations per forward
([Bc, N, C] = 2.6-31 MB each, ≈2.3 GB per step at Bc = 2, and a ring all-reduce moves that twice). Sharding the PIXEL ROWS
of every sample instead keeps every linear layer, LayerNorm, cross-attention, residual add and activation purely local
(weights are replicated: 5.1 GB per GPU of 288 GB) and leaves four small exchanges:

  * self-attention: all-gather of the local K | V rows ([Bc, N/tp, 2C] 16-bit per rank, 70 per forward, ≈0.35 GB per
    step at Bc = 2) — issued right after the K|V projection and overlapped with the Q projection
  * 3x3 convolutions: one halo row above and below the local slab — a neighbour-only exchange (Comm.halo_exchange: one launch over
    the hipIpc peer buffers with IpcComm, an all-gather of the edge rows on other backends), packed with the slab by sx_halo_pack
  * GroupNorm: all-reduce of the fp64 per-(sample, group) sum / sum-of-squares ([Bc, 32, 2] doubles)
  * the final eps rows → all-gather in front of the (replicated) CFG + Euler update

It also has no divisibility constraints beyond H % tp == 0 (SDXL's 10 / 20 attention heads do not divide 4 or 8, and
channel slices of 320/tp are not multiples of the 64-wide k-tiles), so tp ∈ {2, 4, 8} and the edit pipeline's Bc = 3 all run.

Everything here is pure tensor plumbing on whatever device the tensors live on (the gloo CPU tests exercise it).
"""
import torch


def local_rows(x, rank, tp, H, W):
    """x: [B, H*W, C] (replicated) → this rank's slab [B, (H/tp)*W, C] (contiguous copy)."""
    assert H % tp == 0, f"image rows {H} must divide by the sharding degree {tp}"
    B, HW, C = x.shape
    hl = H // tp
    return x.view(B, H, W, C)[:, rank * hl:(rank + 1) * hl].reshape(B, hl * W, C).contiguous()


def gather_rows(x_l, comm):
    """[B, HWl, C] per rank → [B, tp*HWl, C] on every rank (rows in rank order = image order)."""
    g = comm.all_gather(x_l.contiguous())                                   # [tp, B, HWl, C]
    tp, B, HWl, C = g.shape
    return g.permute(1, 0, 2, 3).reshape(B, tp * HWl, C).contiguous()


def with_halo(x_l, comm, Hl, W, left_col=False, bottom=True):
    """x_l: [B, Hl*W, C] local slab → [B, Hl + 1 (+1), W (+1), C]: one row of the neighbouring ranks above (and below);
    zeros at the image border. ``left_col`` adds a zero column on the left (stride-2 convolutions, see unet.py)."""
    B, _, C = x_l.shape
    x4 = x_l.view(B, Hl, W, C)
    edges = torch.stack([x4[:, 0], x4[:, -1]], dim=0).contiguous()          # [2, B, W, C]: my first and last row
    nb = comm.halo_exchange(edges)                                          # [2, B, W, C]: last row of rank - 1, first row of rank + 1
    r, tp = comm.rank, comm.world
    rows = Hl + 1 + (1 if bottom else 0)
    off = 1 if left_col else 0
    if x_l.is_cuda and x_l.element_size() == 2 and C % 8 == 0:
        # slab + neighbour rows + zero borders in ONE launch (sx_halo_pack)
        from . import _lib
        out = torch.empty((B, rows, W + off, C), dtype=x_l.dtype, device=x_l.device)
        lib = _lib.load()
        prev = nb[0].contiguous() if r > 0 else None
        nxt = nb[1].contiguous() if (bottom and r < tp - 1) else None
        _lib.check(lib.sx_halo_pack(x4.contiguous().data_ptr(), prev.data_ptr() if prev is not None else None,
                                    nxt.data_ptr() if nxt is not None else None, out.data_ptr(), B, Hl, W, C, off, 1 if bottom else 0,
                                    torch.cuda.current_stream().cuda_stream), "sx_halo_pack")
        return out
    out = torch.zeros((B, rows, W + off, C), dtype=x_l.dtype, device=x_l.device)
    out[:, 1:Hl + 1, off:] = x4
    if r > 0:
        out[:, 0, off:] = nb[0]
    if bottom and r < tp - 1:
        out[:, Hl + 1, off:] = nb[1]
    return out


def gather_kv(kv_l, comm):
    """kv_l: [B, Nl, 2, h, d] local keys|values → [B, tp*Nl, 2, h, d]."""
    g = comm.all_gather(kv_l.contiguous())
    tp, B, Nl = g.shape[:3]
    return g.permute(1, 0, 2, 3, 4, 5).reshape(B, tp * Nl, *g.shape[3:]).contiguous()
