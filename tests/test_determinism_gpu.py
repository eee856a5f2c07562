"""Run-to-run bit reproducibility. Every 16-bit rounding is a discontinuity: a last-bit difference in a
GroupNorm statistic flips a few roundings behind it, the flips multiply through the layers, and two runs of the same UNet forward end
6e-4 apart — the size of the whole parity budget (profiles/r6_determinism*.log). So every reduction has to add in a fixed order:
the GroupNorm statistics (stand-alone kernel and GEMM / conv epilogue) reduce inside a workgroup through per-thread LDS slots in
index order, and meet across workgroups as fp64 sums of fp32-valued partials (exact, hence order-independent)."""
import pytest
import torch

from oracle import restated_unet as ru, weights

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,HW,C", [(32, 1024, 1280), (8, 16384, 320), (4, 4096, 640)])
def test_groupnorm_statistics_bit_reproducible(dev, B, HW, C):
    from seedx_amd import ops
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(B, HW, C, generator=g) * 1.5 + 0.3).to(dev)
    ga, be = torch.randn(C, generator=g).to(dev), torch.randn(C, generator=g).to(dev)
    ys = [ops.groupnorm(x, ga, be, 32, 1e-5, True, torch.float16).clone() for _ in range(4)]
    for y in ys[1:]:
        assert torch.equal(y, ys[0])


@pytest.mark.parametrize("N,K,B,HW", [(640, 640, 32, 4096), (1280, 1280, 32, 1024), (320, 960, 8, 16384)])
def test_fused_groupnorm_statistics_bit_reproducible(dev, N, K, B, HW):
    """sx_gemm_gn: the statistics a GEMM epilogue accumulates are the same bits in every launch (256x256 and 256x320 tiles, group
    boundaries inside a lane's 4-column piece at C = 320)."""
    from seedx_amd import ops
    g = torch.Generator().manual_seed(6)
    a = (torch.randn(B * HW, K, generator=g) * 0.5).to(torch.float16).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.float16).to(dev)
    bias, res = torch.randn(N, generator=g).to(dev), torch.randn(B * HW, N, generator=g).to(dev)
    bufs = []
    for _ in range(4):
        gs = ops.GnStats(ops.GnStats.arena(1, B, 32, dev)[0], 32, HW)
        ops.gemm(a, w, bias=bias, residual=res, out_dtype=torch.float32, gn=gs)
        if not gs.ready:
            pytest.skip("shape does not run on a ping-pong tile here")
        bufs.append(gs.buf.clone())
    for b_ in bufs[1:]:
        assert torch.equal(b_, bufs[0])


@pytest.mark.parametrize("in_ch", [4, 8])
def test_unet_forward_and_graph_replays_bit_reproducible(dev, in_ch):
    """Three eager forwards of the (miniature) UNet on the same inputs and two complete graph-replayed denoise loops are torch.equal —
    with the statistics fused into the GEMM / conv epilogues and the LayerNorms folded (the defaults). The full-size counterpart is
    tools/determinism_probe.py (profiles/r6_determinism_after.log: 32 samples at 128x128, one forward and 10-step loops, equal)."""
    from seedx_amd.detokenizer import EulerDiscreteScheduler, _DenoiseLoop
    from seedx_amd.unet import UNet2DConditionModel
    cfg = dict(ru.MINI_UNET, in_channels=in_ch)
    sd = ru.unet_sd(cfg)
    m = UNet2DConditionModel(**cfg)
    m.load_state_dict(sd)
    m.to(dev, torch.float16)
    g = torch.Generator().manual_seed(7)
    nb = 2 if in_ch == 4 else 3
    x = torch.randn(nb, in_ch, 16, 16, generator=g)
    ehs = torch.randn(nb, 16, cfg["cross_attention_dim"], generator=g)
    te = torch.randn(nb, cfg["pooled_dim"], generator=g)
    tid = torch.tensor([[1024.0, 1024, 0, 0, 1024, 1024]]).repeat(nb, 1)
    add = {"text_embeds": te.to(dev), "time_ids": tid.to(dev)}
    outs = [m(x.to(dev), 500.0, ehs.to(dev), added_cond_kwargs=add).sample.clone() for _ in range(3)]
    assert torch.equal(outs[1], outs[0]) and torch.equal(outs[2], outs[0])
    _, _, init = ru.euler_tables(6)
    lat0 = torch.randn(1, 4, 16, 16, generator=g) * init
    il3 = torch.randn(3, 4, 16, 16, generator=g) if in_ch == 8 else None
    lats = []
    for _ in range(2):
        loop = _DenoiseLoop(m, use_graph=True)
        lats.append(loop.run(0 if in_ch == 4 else 1, lat0, ehs, te, tid, EulerDiscreteScheduler(), 6, 7.5, 1.5, il3).clone())
    assert torch.equal(lats[0], lats[1])
