"""Full-size parity that round 1 left untested: the oracle restatements run in fp32 ON THE GPU as
the checker; the HIP path runs fp16 (the reference scripts' dtype) with the same seeded weights.
  * 8-channel SDXL UNet, Bc = 3 (the edit pipeline's [text, image, uncond] batch) at 128x128 latents
  * a 50-step t2i CFG loop at full size with the drift reported at steps 1 / 10 / 25 / 50
  * the real SDXL VAE decoder at 128x128 latents (1024x1024 px) and the encoder at 1024 px
  * ViT-G width with B = 20 crops (BASELINE config 5) and a 1536-token Llama prefill
Each test prints its rel-L2; full-size fp16 results are asserted against the north-star's 1e-3 itself (round 3: the bounds
used to be 2-8x looser than what was measured), bf16 (the bench dtype, eps = 7.8e-3) against its own stated bounds."""
import pytest
import torch

from oracle import restated, restated_unet as ru, restated_vae as rv, weights

pytestmark = pytest.mark.gpu


def relerr(x, ref):
    x, ref = x.float(), ref.float().to(x.device)
    return ((x - ref).norm() / ref.norm()).item()


def _to(sd, dev):
    return {k: v.to(dev) for k, v in sd.items()}


def _verdict(name, e, tol):
    print(f"{name}: rel-L2 {e:.3e}  [north-star 1e-3: {'PASS' if e < 1e-3 else 'over'}; asserted fp16 bound {tol:g}]")


def test_unet_full_8ch_bc3_forward(dev):
    from seedx_amd.unet import SDXL_BASE_CONFIG, UNet2DConditionModel
    cfg = dict(ru.FULL_UNET, in_channels=8)
    sd = ru.unet_sd(cfg, device=dev)
    g = torch.Generator().manual_seed(12)
    x = torch.randn(3, 8, 128, 128, generator=g)
    ehs = torch.randn(3, 64, 2048, generator=g)
    te = torch.randn(3, 1280, generator=g)
    tid = torch.tensor([[1024.0, 1024, 0, 0, 1024, 1024]] * 3)
    with torch.no_grad():
        ref = ru.unet_forward(sd, cfg, x.to(dev), 481.0, ehs.to(dev), te.to(dev), tid.to(dev))
    m = UNet2DConditionModel(**dict(SDXL_BASE_CONFIG, in_channels=8))
    m.load_state_dict(sd)
    m.to(dev, torch.float16)
    m._pack()
    del sd
    torch.cuda.empty_cache()
    out = m(x.to(dev), 481.0, ehs.to(dev), added_cond_kwargs={"text_embeds": te.to(dev), "time_ids": tid.to(dev)},
            return_dict=False)[0]
    e = relerr(out, ref)
    _verdict("FULL 8-channel SDXL UNet, Bc=3, 128x128 latents, fp16", e, 1e-3)
    assert out.shape == (3, 4, 128, 128) and torch.isfinite(out).all() and e < 1e-3


# (dtype, generations G, kernel chains, bound at step 1, bound at step 50)
#   fp16 (the reference scripts' dtype): the north-star's 1e-3 at every mark
#   bf16 (the bench dtype) at the bench's geometry: G = 2 generations → UNet batch 4 → the step graph forks into two concurrent
#   kernel chains exactly like the timed bench step; bound 1.2e-2 at step 1 (≈1.5 bf16 eps per module), 2.5e-2 at step 50
@pytest.mark.parametrize("dtype,G,chains,tol1,tol50", [(torch.float16, 1, 1, 1e-3, 1e-3), (torch.bfloat16, 2, 2, 1.2e-2, 2.5e-2)])
def test_full_size_50_step_t2i_loop_drift(dev, dtype, G, chains, tol1, tol50):
    """50 Euler steps of the complete UNet at 128x128 latents, CFG 7.5: HIP graph loop vs the oracle loop (fp32 on the
    GPU), latents compared after 1 / 10 / 25 / 50 steps."""
    from seedx_amd.detokenizer import EulerDiscreteScheduler, _DenoiseLoop
    from seedx_amd.unet import SDXL_BASE_CONFIG, UNet2DConditionModel
    cfg = ru.FULL_UNET
    sd = ru.unet_sd(cfg, device=dev)
    g = torch.Generator().manual_seed(21)
    pe, pen = torch.randn(G, 64, 2048, generator=g), torch.randn(G, 64, 2048, generator=g)
    po, pon = torch.randn(G, 1280, generator=g), torch.randn(G, 1280, generator=g)
    tid = torch.tensor([[1024.0, 1024, 0, 0, 1024, 1024]] * G)
    ts, sig, init = ru.euler_tables(50)
    lat0 = torch.randn(G, 4, 128, 128, generator=g) * init
    marks = (1, 10, 25, 50)
    ref = {}
    with torch.no_grad():
        ehs = torch.cat([pen, pe]).to(dev)
        te = torch.cat([pon, po]).to(dev)
        tids = torch.cat([tid, tid]).to(dev)
        lat = lat0.to(dev).clone()
        for i in range(50):                                               # StableDiffusionXLPipeline loop [ext], restated
            s = float(sig[i])
            inp = torch.cat([lat] * 2) / ((s ** 2 + 1) ** 0.5)
            eps = ru.unet_forward(sd, cfg, inp, ts[i].to(dev), ehs, te, tids)
            eu, et = eps.chunk(2)
            lat = lat + (eu + 7.5 * (et - eu)) * (float(sig[i + 1]) - s)
            if i + 1 in marks:
                ref[i + 1] = lat.clone()
    m = UNet2DConditionModel(**SDXL_BASE_CONFIG)
    m.load_state_dict(sd)
    m.to(dev, dtype)
    m._pack()
    del sd
    torch.cuda.empty_cache()
    trace = {k: None for k in marks}
    loop = _DenoiseLoop(m, use_graph=True)
    loop.chains = chains
    out = loop.run(0, lat0.clone(), torch.cat([pen, pe]), torch.cat([pon, po]), torch.cat([tid, tid]),
                   EulerDiscreteScheduler(), 50, 7.5, trace=trace)
    errs = {k: relerr(trace[k], ref[k]) for k in marks}
    print(f"full-size 50-step t2i loop, {dtype}, {G} generation(s), {chains} kernel chain(s), latents rel-L2 vs fp32 oracle "
          "after N steps: " + ", ".join(f"{k}: {v:.2e}" for k, v in errs.items()))
    assert torch.isfinite(out).all() and relerr(out, ref[50]) == errs[50]
    assert errs[1] < tol1 and max(errs.values()) < tol50                 # no blow-up over 50 steps


def test_vae_full_config_1024px(dev):
    """Real SDXL VAE config (decoder 49.49 M parameters) at 128x128 latents → 1024x1024 image, and the encoder at 1024 px."""
    from seedx_amd.vae import AutoencoderKL
    A = rv.FULL_VAE
    sd_d, sd_e = rv.vae_sd(A, device=dev), rv.vae_encoder_sd(A, device=dev)
    g = torch.Generator().manual_seed(31)
    z = torch.randn(1, 4, 128, 128, generator=g)
    img = torch.rand(1, 3, 1024, 1024, generator=g) * 2 - 1
    h16 = lambda sd: {k: (v.to(torch.float16).to(v.dtype) if v.is_floating_point() else v) for k, v in sd.items()}
    with torch.no_grad():
        # the fp16 VAE's parameters are fp16 values (`.to(dtype=torch.float16)`, eval_text2img_seed_x_i.py:61; upcast_vae only widens
        # them): the oracle is evaluated on those for the fp16 rows, on the fp32 values for bf16's 2e-2
        ref16, ref16_e = rv.vae_decode(h16(sd_d), A, z.to(dev)), rv.vae_encode_mode(h16(sd_e), A, img.to(dev))
        ref32, ref32_e = rv.vae_decode(sd_d, A, z.to(dev)), rv.vae_encode_mode(sd_e, A, img.to(dev))
    # fp16 + force_upcast = what the reference does (its pipeline upcasts the VAE to fp32): fp32-grade operand planes
    for dt, prec, tol in ((torch.float16, None, 1e-4), (torch.float16, "fast", 3e-3), (torch.bfloat16, None, 2e-2)):
        ref, ref_e = (ref16, ref16_e) if dt == torch.float16 else (ref32, ref32_e)
        vae = AutoencoderKL(block_out_channels=A["block_out_channels"], layers_per_block=A["layers_per_block"])
        vae.load_state_dict(dict(sd_d, **sd_e))
        vae.to(dev, dt, precision=prec)
        out = vae.decode(z.to(dev), return_dict=False)[0]
        e = relerr(out, ref)
        dt = f"{dt} precision={prec or 'auto'} (fp32-grade={vae.split})"
        _verdict(f"SDXL VAE decode 128x128 latents -> 1024 px, {dt}", e, tol)
        assert out.shape == (1, 3, 1024, 1024) and torch.isfinite(out).all() and e < tol
        enc = vae.encode(img.to(dev)).latent_dist.mode()
        e = relerr(enc, ref_e)
        _verdict(f"SDXL VAE encode 1024 px -> 128x128 latents, {dt}", e, tol)
        assert enc.shape == (1, 4, 128, 128) and e < tol
        del vae
        torch.cuda.empty_cache()


def test_vit_b20_and_long_prefill(dev):
    """BASELINE config 5 sizes: 20 crops through the full-width ViT (2 of 48 layers) and a 1536-token Llama prefill
    (2 of 40 layers, full dims)."""
    from seedx_amd.llama import LlamaForCausalLM
    from seedx_amd.visual_encoder import VisionTransformerWithAttnPool
    cfg = dict(weights.FULL_VIT, layers=2)
    sd = weights.vit_sd(cfg)
    x = torch.randn(20, 3, 448, 448, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        ref = restated.vit_forward(_to(sd, dev), cfg, x.to(dev))
    m = VisionTransformerWithAttnPool(**cfg)
    m.load_state_dict(sd)
    m.eval().to(dev, dtype=torch.float16)
    e = relerr(m(x), ref)
    _verdict("ViT-G width, B = 20 crops, fp16", e, 1e-3)
    assert e < 1e-3
    lcfg = dict(weights.FULL_LLM, num_hidden_layers=2)
    lsd = weights.llama_sd(lcfg)
    xe = torch.randn(1, 1536, 5120, generator=torch.Generator().manual_seed(6)) * 0.5
    dt = torch.float16
    with torch.no_grad():
        lref, _, href = restated.llama_forward(_to(lsd, dev), lcfg, xe.to(dev), table_dtype=dt)
    llm = LlamaForCausalLM(dict(lcfg), max_cache_len=2048)
    llm.load_state_dict(lsd)
    llm.eval().to(dev, dt)
    out = llm(inputs_embeds=xe.to(dev), output_hidden_states=True)
    e_l, e_h = relerr(out["logits"][0, -1], lref[0, -1]), relerr(out["hidden_states"][-1], href)
    _verdict("Llama-13B dims, 1536-token prefill, fp16: last-position logits", e_l, 1e-3)
    _verdict("Llama-13B dims, 1536-token prefill, fp16: final-norm states", e_h, 1e-3)
    assert e_l < 1e-3 and e_h < 1e-3


def test_unet_layernorm_fold_full_size(dev, monkeypatch):
    """The complete SDXL UNet at 16 samples (the batch where every transformer GEMM runs on a ping-pong tile) with the transformer
    blocks' LayerNorms folded into their neighbour GEMMs (ops.LnRows, sx_gemm_ln) against the unfolded launches on the same
    weights, and both against the fp32 oracle on the first two samples (samples are independent through the whole UNet)."""
    from seedx_amd import unet as U
    cfg = ru.FULL_UNET
    sd = ru.unet_sd(cfg, device=dev)
    g = torch.Generator().manual_seed(5)
    B = 16
    x = torch.randn(B, 4, 128, 128, generator=g)
    ehs = torch.randn(B, 64, 2048, generator=g)
    te = torch.randn(B, 1280, generator=g)
    tid = torch.tensor([[1024.0, 1024, 0, 0, 1024, 1024]] * B)
    with torch.no_grad():
        ref = ru.unet_forward(sd, cfg, x[:2].to(dev), 981.0, ehs[:2].to(dev), te[:2].to(dev), tid[:2].to(dev))
    monkeypatch.setattr(U, "LN_FOLD", True)
    m = U.UNet2DConditionModel(**U.SDXL_BASE_CONFIG)
    m.load_state_dict(sd)
    m.to(dev, torch.float16)
    m._pack()
    del sd
    torch.cuda.empty_cache()
    assert m._ln_fold_ok(B * 1024, 1280) and m._ln_fold_ok(B * 4096, 640)
    kw = dict(added_cond_kwargs={"text_embeds": te.to(dev), "time_ids": tid.to(dev)}, return_dict=False)
    from seedx_amd import _lib
    lib = _lib.load()
    calls = {"n": 0}
    real = lib.sx_gemm_ln

    def counted(*a):
        calls["n"] += 1
        return real(*a)
    monkeypatch.setattr(lib, "sx_gemm_ln", counted)
    out_f = m(x.to(dev), 981.0, ehs.to(dev), **kw)[0]
    n_fold = calls["n"]
    monkeypatch.setattr(U, "LN_FOLD", False)
    out_u = m(x.to(dev), 981.0, ehs.to(dev), **kw)[0]
    assert calls["n"] == n_fold and n_fold == 2 * 3 * 70     # 70 blocks x 3 norms, a producer and a consumer launch each
    e_f, e_u, d = relerr(out_f[:2], ref), relerr(out_u[:2], ref), relerr(out_f, out_u)
    print(f"FULL SDXL UNet, 16 samples, fp16: folded LayerNorms rel-L2 {e_f:.3e}, separate LayerNorm launches {e_u:.3e}, "
          f"folded vs separate {d:.3e}")
    assert torch.isfinite(out_f).all() and e_f < 1e-3 and e_u < 1e-3
