"""Parity at BASELINE.json's FULL dimensions AND FULL DEPTH: the complete 48-layer ViT-G, the complete
40-layer Llama-2-13B-dim decoder (prefill + 128 decode steps), and ONE config-0 generation end to end
(image → ViT → input resampler → LLM greedy decode with an image block → output resampler → ResamplerXLV2 → 5 UNet CFG steps →
VAE decode). The checker is the restated oracle (oracle/restated*.py: pinned against the reference's own modules at mini
dimensions) executed in fp32 ON THE GPU with the same seeded fp32 weights; the HIP path runs fp16, the reference scripts' dtype
(eval_img2text_seed_x_i.py:60-62) and the dtype north_star's 1e-3 (rel-L2 vs the fp32 oracle, SURVEY.md §7 hard part 5) is
stated for. Reference: modeling_llama_xformer.py:554-599 (40 decoder layers + final norm), qwen_visual.py:300-317 (48 blocks).

Weights come from seedx_amd/synthetic.py's GPU-side generators in fp32 (13 B parameters are not practical to draw on the host):
same key names as the reference's state dicts, so the one dict feeds the oracle as is and the HIP modules through load_state_dict.
The LLM's weights are rounded to fp16-representable values first — a real SEED-X checkpoint IS 16-bit (llm_seed_x_i.yaml: torch_dtype fp16),
so oracle and HIP path see the same weights ("same inputs"); round 4 measured the un-rounded variant too (2.26e-3 vs 2.18e-3 then).
Round 5: the decoder runs its default precise mode (fp32-grade activations, csrc/precise.hip) and the asserted bound is north_star's 1e-3.
Memory: ≈ 52 GB (fp32 LLM) + 26 GB (fp16 LLM) + 8 + 4 (ViT) + 10 + 5 (UNet) of the 288 GB."""
import math

import pytest
import torch

from oracle import restated, restated_adapter as ra, restated_unet as ru, restated_vae as rv, weights

pytestmark = pytest.mark.gpu
DT = torch.float16


def relerr(x, ref):
    x, ref = x.float(), ref.float().to(x.device)
    return ((x - ref).norm() / ref.norm()).item()


def _ckpt16(sd):
    """The weights a 16-bit checkpoint holds (the reference's scripts load every module with `.to(dtype=torch.float16)`,
    eval_img2text_seed_x_i.py:60-62, eval_text2img_seed_x_i.py:16-19): oracle and HIP path then see the SAME weights. The full-size tests of
    tests/test_fullsize*_gpu.py keep un-rounded fp32 weights (and hold 1e-3 with that handicap)."""
    for k in sd:
        if torch.is_floating_point(sd[k]):
            sd[k] = sd[k].to(DT).float()
    return sd


def _report(name, e, bound):
    print(f"[full depth] {name}: rel-L2 {e:.3e}  (north-star 1e-3: {'met' if e < 1e-3 else 'NOT met'}; asserted < {bound:g})")


@pytest.fixture(scope="module")
def vit48(dev):
    from seedx_amd import synthetic as syn
    from seedx_amd.visual_encoder import VisionTransformerWithAttnPool
    cfg = dict(weights.FULL_VIT)
    sd = _ckpt16(syn.vit_state_dict(cfg, dev, torch.float32))
    sd["attn_pool.pos_embed"] = restated.sincos_2d(cfg["output_dim"], int(math.sqrt(cfg["n_queries"]))).to(dev)
    m = VisionTransformerWithAttnPool(**cfg)
    m.load_state_dict(sd)
    m.eval().to(dev, dtype=DT)
    m._pack()
    return cfg, sd, m


@pytest.fixture(scope="module")
def llm40(dev):
    from seedx_amd import synthetic as syn
    from seedx_amd.llama import LlamaForCausalLM
    cfg = dict(weights.FULL_LLM)
    sd = syn.llama_state_dict(cfg, dev, torch.float32)
    _ckpt16(sd)                                         # what a 16-bit checkpoint stores (in place: 52 GB of fp32 tensors)
    llm = LlamaForCausalLM(dict(cfg), max_cache_len=512)                    # the shipped fp16 default: precise mode with the MIXED KV cache
    assert llm.precise                                                      # (k fp32, v 16-bit); the all-fp32 floor has its own test below
    llm.load_state_dict(sd)
    llm.eval().to(dev, DT)
    llm._pack()
    assert llm.kv_v16 and llm._P["vc"].dtype == DT and llm._P["kc"].dtype == torch.float32
    llm._sd = None
    torch.cuda.empty_cache()
    return cfg, sd, llm


def test_vit_g_48_layers(dev, vit48):
    cfg, sd, m = vit48
    x = torch.randn(2, 3, 448, 448, generator=torch.Generator().manual_seed(0)).to(dev)
    with torch.no_grad():
        ref = restated.vit_forward(sd, cfg, x)
    y = m(x)
    e = relerr(y, ref)
    _report("ViT-G/448, 48 layers, B = 2 crops, fp16: [2, 256, 4096] features", e, 1e-3)
    assert y.shape == (2, 256, 4096) and torch.isfinite(y).all()
    assert e < 1e-3


def test_llama_13b_40_layers_prefill_and_128_decode_steps(dev, llm40):
    """165-token prefill (inputs_embeds) + 128 greedy single-token steps through the GEMV / split-KV decode path. The oracle is
    teacher-forced on OUR tokens in one causal fp32 pass over all 293 positions (exactly what 128 cached steps compute); reported:
    logits rel-L2 at decode steps 1 / 64 / 128, the arg-max agreement rate, and for every disagreement the oracle's own margin
    between its choice and ours in units of its logit spread."""
    cfg, sd, llm = llm40
    V = cfg["vocab_size"]
    xe = (torch.randn(1, 165, 5120, generator=torch.Generator().manual_seed(1)) * 0.5).to(dev)
    out = llm(inputs_embeds=xe, use_cache=True, output_hidden_states=True)
    assert tuple(out.logits.shape) == (1, 165, V)
    ours_prefill_logits, ours_prefill_hidden = out.logits[0].clone(), out.hidden_states[-1][0].clone()
    toks, step_logits = [], []
    nxt = int(out.logits[0, -1].argmax())
    pkv = out.past_key_values
    for _ in range(128):
        toks.append(nxt)
        o = llm(input_ids=torch.tensor([[nxt]]), past_key_values=pkv, use_cache=True, logits_positions="last")
        pkv = o.past_key_values
        step_logits.append(o.logits[0, -1].clone())
        nxt = int(o.logits[0, -1].argmax())
    assert int(llm._P["pos"].item()) == 165 + 128
    with torch.no_grad():
        emb = sd["model.embed_tokens.weight"][torch.tensor(toks, device=dev)].unsqueeze(0)
        lref, _, href = restated.llama_forward(sd, cfg, torch.cat([xe, emb], dim=1), None, table_dtype=DT)
    e_pl, e_ph = relerr(ours_prefill_logits, lref[0, :165]), relerr(ours_prefill_hidden, href[0, :165])
    # yardstick on the same prefill: the reference's own 16-bit dtype flow (16-bit residual stream and module outputs,
    # restated.llama_forward_16bit_like_reference) against the same fp32 oracle
    with torch.no_grad():
        lrl, _ = restated.llama_forward_16bit_like_reference(sd, cfg, xe, DT)
    e_reflike = relerr(lrl[0], lref[0, :165])
    BOUND = 1e-3
    _report("Llama-13B dims, 40 layers (precise mode, mixed KV cache = the fp16 default), 165-token prefill: logits of all positions", e_pl, BOUND)
    _report("Llama-13B dims, 40 layers, 165-token prefill: final-norm states", e_ph, BOUND)
    print(f"[full depth]   yardstick: the reference's own fp16 dtype flow vs the fp32 oracle: rel-L2 {e_reflike:.3e} "
          f"(HIP path / reference-like = {e_pl / e_reflike:.2f})")
    errs = {k: relerr(step_logits[k - 1], lref[0, 164 + k]) for k in (1, 64, 128)}
    for k, e in errs.items():
        _report(f"Llama-13B dims, 40 layers, decode step {k}: logits", e, BOUND)
    # arg-max agreement: our token k+1 was picked from OUR logits at step k; the oracle saw the same inputs
    agree, worst = 0, 0.0
    for k in range(128):
        ref_l = lref[0, 164 + k]
        ours_l = ours_prefill_logits[-1] if k == 0 else step_logits[k - 1]
        o_arg, m_arg = int(ref_l.argmax()), int(ours_l.argmax())
        assert m_arg == toks[k]
        if o_arg == m_arg:
            agree += 1
        else:
            margin = float(ref_l[o_arg] - ref_l[m_arg]) / float(ref_l.std())
            worst = max(worst, margin)
            print(f"  step {k}: oracle arg-max {o_arg} vs ours {m_arg}; oracle margin {margin:.2e} of its logit std")
    print(f"[full depth] greedy arg-max agreement over 128 teacher-forced steps: {agree}/128; worst margin of a disagreement "
          f"{worst:.2e} logit-std")
    # Round 4 (one 16-bit rounding per MFMA operand): 2.3e-3 at 40 layers. Round 5 (precise mode: two operand planes, fp32 q / k / v /
    # cache / attention): north_star's 1e-3 is the asserted bound for prefill logits, states and decode steps 1 / 64 / 128
    assert max(e_pl, e_ph, *errs.values()) < BOUND
    assert e_pl <= 0.5 * e_reflike
    assert agree >= 126 and worst < 2e-3          # a disagreement is only acceptable inside the noise of a near-tie


def test_llama_13b_40_layers_all_fp32_cache_floor(dev, llm40):
    """The precise mode's parity FLOOR: `kv_v16=False` (k and v fp32, round 5's mode; the default for bf16 models) at full depth — prefill
    logits of all positions and 16 cached decode steps against the fp32 oracle: 2.9e-5 / 2.0e-5. VERDICT r5 item 1(b) asked for the
    mixed cache (v 16-bit, three quarters of the KV bytes) to be measured at 40 layers and shipped if <= 7e-4: it measures 5.6e-4 /
    5.9e-4 (the test above, which now runs the shipped default) against the budget's predicted 0.59e-3 (tools/llm_error_budget.py)."""
    from seedx_amd.llama import LlamaForCausalLM
    cfg, sd, _ = llm40
    llm = LlamaForCausalLM(dict(cfg), max_cache_len=256, kv_v16=False)
    llm.load_state_dict(sd)
    llm.eval().to(dev, DT)
    llm._pack()
    assert llm.precise and not llm.kv_v16 and llm._P["vc"].dtype == torch.float32
    xe = (torch.randn(1, 165, 5120, generator=torch.Generator().manual_seed(1)) * 0.5).to(dev)
    out = llm(inputs_embeds=xe, use_cache=True)
    ours = out.logits[0].clone()
    toks, step_logits = [], []
    nxt, pkv = int(ours[-1].argmax()), out.past_key_values
    for _ in range(16):
        toks.append(nxt)
        o = llm(input_ids=torch.tensor([[nxt]]), past_key_values=pkv, use_cache=True, logits_positions="last")
        pkv = o.past_key_values
        step_logits.append(o.logits[0, -1].clone())
        nxt = int(o.logits[0, -1].argmax())
    with torch.no_grad():
        emb = sd["model.embed_tokens.weight"][torch.tensor(toks, device=dev)].unsqueeze(0)
        lref, _, _ = restated.llama_forward(sd, cfg, torch.cat([xe, emb], dim=1), None, table_dtype=DT)
    e_pl = relerr(ours, lref[0, :165])
    e_dec = max(relerr(step_logits[k], lref[0, 165 + k]) for k in range(16))
    _report("Llama-13B dims, 40 layers, precise mode with the ALL-fp32 cache (kv_v16=False): prefill logits of all positions", e_pl, 1e-4)
    _report("Llama-13B dims, 40 layers, ALL-fp32 cache: worst of 16 cached decode steps", e_dec, 1e-4)
    del llm
    torch.cuda.empty_cache()
    assert max(e_pl, e_dec) < 1e-4


def test_llama_13b_40_layers_plain16_flow(dev, llm40):
    """The plain 16-bit flow (`precise=False` / SX_LLM_PRECISE=0: one 16-bit rounding per MFMA operand, 16-bit KV cache) at FULL depth — the
    companion run of BASELINE config 2 (`value_plain16_batch32`) and the plain-flow row of bench.py's PARITY_BOUND: all-position prefill logits and 8 cached decode steps of
    the 40-layer decoder against the fp32 oracle, asserted at 3e-3 (measured 2.3e-3 in round 4; north_star's 1e-3 is met by the
    precise mode only — the test above)."""
    from seedx_amd.llama import LlamaForCausalLM
    cfg, sd, _ = llm40
    llm = LlamaForCausalLM(dict(cfg), max_cache_len=256, max_batch=32, precise=False)
    assert not llm.precise
    llm.load_state_dict(sd)
    llm.eval().to(dev, DT)
    llm._pack()
    xe = (torch.randn(1, 165, 5120, generator=torch.Generator().manual_seed(1)) * 0.5).to(dev)
    out = llm(inputs_embeds=xe, use_cache=True)
    ours = out.logits[0].clone()
    toks, step_logits = [], []
    nxt, pkv = int(ours[-1].argmax()), out.past_key_values
    for _ in range(8):
        toks.append(nxt)
        o = llm(input_ids=torch.tensor([[nxt]]), past_key_values=pkv, use_cache=True, logits_positions="last")
        pkv = o.past_key_values
        step_logits.append(o.logits[0, -1].clone())
        nxt = int(o.logits[0, -1].argmax())
    with torch.no_grad():
        emb = sd["model.embed_tokens.weight"][torch.tensor(toks, device=dev)].unsqueeze(0)
        lref, _, _ = restated.llama_forward(sd, cfg, torch.cat([xe, emb], dim=1), None, table_dtype=DT)
    e_pl = relerr(ours, lref[0, :165])
    e_dec = max(relerr(step_logits[k], lref[0, 165 + k]) for k in range(8))
    _report("Llama-13B dims, 40 layers, PLAIN 16-bit flow (max_batch 32): prefill logits of all positions", e_pl, 3e-3)
    _report("Llama-13B dims, 40 layers, PLAIN 16-bit flow: worst of 8 cached decode steps", e_dec, 3e-3)
    del llm
    torch.cuda.empty_cache()
    assert max(e_pl, e_dec) < 3e-3


def test_config0_one_generation_end_to_end(dev, vit48, llm40):
    """BASELINE config 0 at full size, one request: uint8 image → GPU preprocessing → ViT-G (48) → input resampler → 165-token
    prefill → 4 text tokens + <img> + 64 forced + </img> → output resampler → ResamplerXLV2 → 50 CFG-7.5 Euler steps of the
    complete SDXL UNet @128² → SDXL VAE decode (fp32-grade) → [1, 3, 1024, 1024]. The oracle chain runs on ITS OWN intermediate
    results (errors compound stage to stage), teacher-forced only on the token ids."""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from seedx_amd import synthetic as syn
    from seedx_amd.detokenizer import EulerDiscreteScheduler, ResamplerXLV2, SDXLAdapter
    from seedx_amd.seed_x import ContinuousLVLM
    from seedx_amd.unet import SDXL_BASE_CONFIG, UNet2DConditionModel
    from seedx_amd.vae import AutoencoderKL
    from seedx_amd.visual_encoder import Resampler
    vcfg, sd_vit, vit = vit48
    lcfg, sd_llm, llm = llm40
    H = lcfg["hidden_size"]
    tok = bench.BenchTokenizer()
    # ---- HIP path ----------------------------------------------------------------------------------------------------
    sd_agent = _ckpt16(syn.agent_state_dict(H, 4096, dev, torch.float32))
    sd_agent["input_resampler.pos_embed"] = restated.sincos_2d(H, 8).to(dev)
    sd_agent["output_resampler.pos_embed"] = restated.sincos_2d(4096, 8).to(dev)
    agent = ContinuousLVLM(llm, Resampler(8, H, 32, kv_dim=4096), Resampler(8, 4096, 32, kv_dim=H), add_patch_pos=True, vit_down=True)
    agent.load_state_dict(sd_agent)
    agent.eval().to(dev, DT)
    images, ids = bench.make_inputs(dev)
    crops, ppos = bench.preprocess(images, dev)                                  # [2, 3, 448, 448] fp32, bit-exact vs the reference
    emb = vit(crops)
    from seedx_amd import image_ops
    mask = image_ops.marker_mask(torch.tensor(ids, dtype=torch.long, device=dev), tok.BOI, tok.EOI, tok.BOP, tok.EOP).view(1, -1)
    n_text = 4
    out = agent.generate(tok, input_ids=[ids], image_embeds=emb, embeds_cmp_mask=torch.tensor([True, True]), ids_cmp_mask=mask,
                         patch_positions=ppos, max_new_tokens=n_text + 66 + 1, eos_token_id=None, force_image_at=n_text)
    new = out["generate_ids"].tolist()
    assert out["has_img_output"] and tuple(out["img_gen_feat"].shape) == (1, 64, 4096)
    ucfg = ru.FULL_UNET
    sd_u = _ckpt16(ru.unet_sd(ucfg, device=dev))
    sd_x = _ckpt16(syn.xlv2_state_dict(weights.FULL_XLV2, dev, torch.float32))
    unet = UNet2DConditionModel(**SDXL_BASE_CONFIG)
    unet.load_state_dict(sd_u)
    rs = ResamplerXLV2(normalize=False, **weights.FULL_XLV2)
    rs.load_state_dict(sd_x, prefix="resampler.")
    A = rv.FULL_VAE
    sd_vae = _ckpt16(rv.vae_sd(A, device=dev))
    vae = AutoencoderKL(block_out_channels=A["block_out_channels"], layers_per_block=A["layers_per_block"])
    vae.load_state_dict(dict(sd_vae, **_ckpt16(rv.vae_encoder_sd(A, device=dev))))
    vae.to(dev, DT)                                   # fp16 + force_upcast → the fp32-grade mode, as the reference upcasts (pipeline…:967-970)
    ad = SDXLAdapter(unet, rs, vit_down=True)
    ad.init_pipe(vae=vae, scheduler=EulerDiscreteScheduler(), visual_encoder=vit, image_transform=None, discrete_model=None,
                 dtype=DT, device=dev)
    noise = torch.randn(1, 4, 128, 128, generator=torch.Generator().manual_seed(42))
    lat = ad.generate(image_embeds=out["img_gen_feat"], latents=noise.clone(), num_inference_steps=N_STEPS, output_type="latent")
    img = ad.generate(image_embeds=out["img_gen_feat"], latents=noise.clone(), num_inference_steps=N_STEPS, output_type="pt")
    # ---- oracle chain (fp32 on the GPU) --------------------------------------------------------------------------------
    with torch.no_grad():
        ref_emb = restated.vit_forward(sd_vit, vcfg, crops)
        img_ids = tok.encode("".join(["<img>"] + [f"<img_{i:05d}>" for i in range(64)] + ["</img>"]))
        old_device = torch.get_default_device()
        torch.set_default_device(dev)                # restated.lvlm_generate builds its small index / mask / RoPE tensors on the default device
        try:
            ref = restated.lvlm_generate(sd_llm, sd_agent, lcfg, {"in_heads": 32, "out_heads": 32}, ids, ref_emb,
                                         torch.tensor([True, True]), mask, ppos.to(dev), img_ids, tok.BOI, tok.EOI, len(new), 64,
                                         None, DT, new, [])
        finally:
            torch.set_default_device(old_device)
        ref_lat = ra.adapter_generate(sd_vit, vcfg, sd_x, weights.FULL_XLV2, sd_u, ucfg, noise.to(dev), N_STEPS,
                                      image_embeds=ref["img_gen_feat"])
        ref_img = ra.decode_to_pt(sd_vae, A, ref_lat)
        # the same oracle stages fed with the HIP path's OWN stage inputs: north_star's "within 1e-3 of the reference on the same inputs"
        # per module (the chain above compounds the stages' errors on top of each other)
        torch.set_default_device(dev)
        try:
            same = restated.lvlm_generate(sd_llm, sd_agent, lcfg, {"in_heads": 32, "out_heads": 32}, ids, emb.float(),
                                          torch.tensor([True, True]), mask, ppos.to(dev), img_ids, tok.BOI, tok.EOI, len(new), 64,
                                          None, DT, new, [])
        finally:
            torch.set_default_device(old_device)
        same_lat = ra.adapter_generate(sd_vit, vcfg, sd_x, weights.FULL_XLV2, sd_u, ucfg, noise.to(dev), N_STEPS,
                                       image_embeds=out["img_gen_feat"].float())
        same_img = ra.decode_to_pt(sd_vae, A, lat.float())
    chain = {"ViT features [2,256,4096]": relerr(emb, ref_emb),
             "LLM final-norm states of the 70 fed tokens": relerr(out["last_hidden_states"], ref["last_hidden"]),
             "output-resampled image features [1,64,4096]": relerr(out["img_gen_feat"], ref["img_gen_feat"]),
             "latents after the 50 UNet CFG steps [1,4,128,128]": relerr(lat, ref_lat),
             "decoded image in [0,1] [1,3,1024,1024]": relerr(img, ref_img)}
    stage = {"ViT-G (48 layers) on the same crops": chain["ViT features [2,256,4096]"],
             "input resampler + LLM (40 layers, 70 fed tokens) on the same ViT features: final-norm states":
                 relerr(out["last_hidden_states"], same["last_hidden"]),
             "... + output resampler: image features": relerr(out["img_gen_feat"], same["img_gen_feat"]),
             "ResamplerXLV2 + 50 UNet CFG-7.5 Euler steps on the same image features: latents": relerr(lat, same_lat),
             "VAE decode of the same latents: image": relerr(img, same_img)}
    for k, e in chain.items():
        _report("config-0 chain (oracle on its OWN intermediates: errors compound), " + k, e, CHAIN_BOUND)
    for k, e in stage.items():
        _report("config-0 stage on the SAME inputs, " + k, e, 1e-3)
    assert tuple(img.shape) == (1, 3, 1024, 1024) and torch.isfinite(img).all()
    assert max(stage.values()) < 1e-3
    assert max(chain.values()) < CHAIN_BOUND


CHAIN_BOUND = 2.5e-3      # five stages of <= 1e-3 each on top of each other (round 4 measured 2.05e-3 with the 16-bit LLM flow)
N_STEPS = 50              # BASELINE's de-tokenizer setting (eval_text2img_seed_x_i.py:91); round 4 ran 5 steps here — with 5 large Euler
                          # steps the CFG-amplified (x 7.5) UNet error weighs 1.5e-3 on the latents, at the real 50 it is < 1e-3
