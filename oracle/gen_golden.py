"""Generate the golden fixtures in tests/golden/ from the REFERENCE's own modules (build container only).

The reference ships no golden vectors, so the fixtures are produced here by importing
``/root/reference/src/models/*`` on CPU (through oracle/refshim.py), loading the seeded weights of oracle/weights.py
(regenerated deterministically from the seed on any machine with this torch build) and recording input/output
tensors at "mini" dimensions. They pin (a) oracle/restated.py in the CPU suite and (b) the HIP path in the GPU suite
on machines where /root/reference does not exist.

    python -m oracle.gen_golden          # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import refshim, weights  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def _save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name), **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                                     for k, v in arrs.items()})
    print("wrote", name, {k: tuple(np.asarray(v.detach().cpu() if torch.is_tensor(v) else v).shape) for k, v in arrs.items()})


def main():
    mods = refshim.reference_modules()
    g = torch.Generator().manual_seed(2024)
    with torch.no_grad():
        # ---- path A: VisionTransformerWithAttnPool (qwen_visual.py) -------------------------------------
        for tag, cfg in (("vit_hd128", weights.MINI_VIT), ("vit_hd104", weights.MINI_VIT_104)):
            m = mods["VisionTransformerWithAttnPool"](**cfg).eval()
            m.load_state_dict(weights.vit_sd(cfg), strict=True)
            x = torch.randn(2, 3, cfg["image_size"], cfg["image_size"], generator=g)
            _save(f"{tag}.npz", x=x, y=m(x).float())
        # ---- path B: LlamaForCausalLM forward (prefill + one cached step) + logits processor ----------------
        from transformers import LlamaConfig
        cfg = weights.MINI_LLM
        sd = weights.llama_sd(cfg)
        llm = mods["LlamaForCausalLM"](LlamaConfig(**cfg)).eval()
        full = dict(llm.state_dict())
        full.update(sd)
        llm.load_state_dict(full, strict=True)
        x = torch.randn(1, 23, cfg["hidden_size"], generator=g) * 0.5
        o = llm(inputs_embeds=x, attention_mask=torch.ones(1, 23, dtype=torch.long), use_cache=True,
                output_hidden_states=True, return_dict=True)
        tok = torch.tensor([[17]])
        o2 = llm(input_ids=tok, attention_mask=torch.ones(1, 24, dtype=torch.long), past_key_values=o.past_key_values,
                 use_cache=True, output_hidden_states=True, return_dict=True)
        _save("llama_mini.npz", x=x, logits=o.logits, hidden=o.hidden_states[-1], tok=tok, logits2=o2.logits,
              hidden2=o2.hidden_states[-1])
        ids = list(range(400, 466))

        class Tok:
            def encode(self, s, add_special_tokens=False):
                return ids
        proc = mods["AutoImageTokenGenerationProcessor"](Tok(), num_img_gen_tokens=64)
        lasts, scores_in, scores_out = [5, 400, 433, 464, 465], [], []
        for last in lasts:
            sc = torch.randn(1, 500, generator=g) - 3.0
            scores_in.append(sc.clone())
            scores_out.append(proc(torch.tensor([[1, 2, last]]), sc.clone()))
        _save("logits_rule.npz", last=np.array(lasts), scores_in=torch.cat(scores_in), scores_out=torch.cat(scores_out))
        # ---- Resampler (LLM side) and ResamplerXLV2 (path C head) --------------------------------------------
        r = mods["Resampler"](grid_size=4, embed_dim=320, num_heads=2, kv_dim=256).eval()
        r.load_state_dict(weights.resampler_sd(weights._g(7), "", 4, 320, 256), strict=True)
        x = torch.randn(2, 16, 256, generator=g)
        _save("resampler_mini.npz", x=x, y=r(x))
        cfgx = weights.MINI_XLV2
        m = mods["ResamplerXLV2"](normalize=False, **cfgx).eval()
        m.load_state_dict(weights.xlv2_sd(cfgx, pre=""), strict=True)
        x = torch.randn(2, 24, cfgx["embedding_dim"], generator=g)
        pe, pooled = m(x)
        _save("xlv2_mini.npz", x=x, prompt=pe, pooled=pooled)


# ---- path C: the reference's OWN adapter front ends + edit pipeline, executed over oracle/diffusers_shim.py ----------
DETOK_VIT, DETOK_XLV2, DETOK_VAE, detok_unet_cfg = weights.DETOK_VIT, weights.DETOK_XLV2, weights.DETOK_VAE, weights.detok_unet_cfg


def build_reference_adapter(edit, with_vae=True):
    """Instantiates the reference's SDXLAdapter / SDXLAdapterWithLatentImage (adapter_modules.py) on CPU/fp32 around
    the reference ViT + ResamplerXLV2 classes and the duck-typed oracle UNet / scheduler / VAE, with the seeded weights
    every test regenerates. The edit variant goes through the reference's own conv_in surgery (set_trainable :186-198)
    and `load_state_dict(ckpt, strict=False)` exactly as `from_pretrained` :211-220 does."""
    from oracle import diffusers_shim as ds, restated_unet as ru, restated_vae as rv
    SDXLAdapter, SDXLAdapterWithLatentImage, _ = ds.reference_adapters()
    mods = refshim.reference_modules()
    vit = mods["VisionTransformerWithAttnPool"](**DETOK_VIT).eval()
    vit.load_state_dict(weights.vit_sd(DETOK_VIT), strict=True)
    rs = mods["ResamplerXLV2"](normalize=False, **DETOK_XLV2).eval()
    sd4 = ru.unet_sd(detok_unet_cfg(4))
    unet = ds.OracleUNet(detok_unet_cfg(4), sd4, sample_size=16)
    cls = SDXLAdapterWithLatentImage if edit else SDXLAdapter
    ad = cls(unet=unet, resampler=rs, vit_down=True)
    ckpt = {k: v for k, v in weights.xlv2_sd(DETOK_XLV2, pre="resampler.").items()}
    ckpt.update({"unet." + k: v for k, v in ru.unet_sd(detok_unet_cfg(8 if edit else 4)).items()})
    missing, unexpected = ad.load_state_dict(ckpt, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    vae = ds.OracleVAE(DETOK_VAE, rv.vae_sd(DETOK_VAE), rv.vae_encoder_sd(DETOK_VAE)) if with_vae else None
    kw = {} if edit else dict(discrete_model=None)
    ad.init_pipe(vae=vae, scheduler=ds.EulerDiscreteScheduler(), visual_encoder=vit, image_transform=None,
                 dtype=torch.float32, device="cpu", **kw)
    return ad.eval()


def detok_inputs():
    g = torch.Generator().manual_seed(4242)
    return dict(image_tensor=torch.randn(1, 3, 112, 112, generator=g), feats=torch.randn(1, 16, 256, generator=g),
                noise=torch.randn(1, 4, 16, 16, generator=g), image_latents=torch.randn(1, 4, 16, 16, generator=g),
                src_image=torch.rand(1, 3, 128, 128, generator=g))


def run_reference_detok():
    """Returns {file: {array name: tensor}} produced by running the reference adapters."""
    inp = detok_inputs()
    out = {}
    with torch.no_grad():
        # -- t2i (SDXLAdapter.generate adapter_modules.py:132-169; loop = restated [ext] StableDiffusionXLPipeline) --
        ad = build_reference_adapter(edit=False)
        pe, pen, po, pon = ad.get_image_embeds(image_tensor=inp["image_tensor"])
        pe2, pen2, po2, pon2 = ad.get_image_embeds(image_embeds=inp["feats"], image_size=112)
        traj = []
        lat = ad.generate(image_embeds=inp["feats"], input_image_size=112, height=128, width=128, num_inference_steps=5,
                          latents=inp["noise"].clone(), output_type="latent",
                          callback=lambda i, t, l: traj.append(l.clone()))
        img = ad.generate(image_tensor=inp["image_tensor"], height=128, width=128, num_inference_steps=2,
                          latents=inp["noise"].clone(), output_type="pt")
        # -- SDXLAdapter.forward (adapter_modules.py:39-52): one UNet forward on noisy latents + the MSE against the noise --
        gf = torch.Generator().manual_seed(515)
        fwd_noisy = torch.randn(2, 4, 16, 16, generator=gf)
        fwd_noise = torch.randn(2, 4, 16, 16, generator=gf)
        fwd_feats = torch.randn(2, 16, 256, generator=gf)
        fwd_t = torch.tensor([981.0, 37.0])
        fwd_tid = torch.tensor([[128.0, 128, 0, 0, 128, 128]] * 2)
        fwd = ad(fwd_noisy, fwd_t, fwd_feats, None, fwd_noise, fwd_tid)
        out["t2i_mini.npz"] = dict(image_tensor=inp["image_tensor"], feats=inp["feats"], noise=inp["noise"],
                                   tensor_prompt=pe, tensor_prompt_neg=pen, tensor_pooled=po, tensor_pooled_neg=pon,
                                   embeds_prompt=pe2, embeds_prompt_neg=pen2, embeds_pooled=po2, embeds_pooled_neg=pon2,
                                   latents_traj=torch.stack(traj), latents=lat, image_pt=img,
                                   fwd_noisy=fwd_noisy, fwd_noise=fwd_noise, fwd_feats=fwd_feats, fwd_t=fwd_t, fwd_time_ids=fwd_tid,
                                   fwd_noise_pred=fwd["noise_pred"], fwd_loss=fwd["total_loss"].reshape(1))
        # -- edit (SDXLAdapterWithLatentImage.generate :249-287 → reference pipeline __call__ :618-994) ---------------
        ad = build_reference_adapter(edit=True)
        traj = []
        lat = ad.generate(image_embeds=inp["feats"], latent_image=inp["image_latents"], input_image_size=112, height=128,
                          width=128, num_inference_steps=5, latents=inp["noise"].clone(), output_type="latent",
                          callback=lambda i, t, l: traj.append(l.clone()))
        lat0 = ad.generate(image_embeds=inp["feats"], latent_image=None, input_image_size=112, height=128, width=128,
                           num_inference_steps=3, latents=inp["noise"].clone(), output_type="latent")
        # source given as an RGB image in [0,1] (what the eval scripts pass, as PIL): preprocess → VAE encode → mode()
        lat_img = ad.generate(image_embeds=inp["feats"], latent_image=inp["src_image"], input_image_size=112, height=128,
                              width=128, num_inference_steps=2, latents=inp["noise"].clone(), output_type="latent")
        img = ad.generate(image_embeds=inp["feats"], latent_image=inp["image_latents"], input_image_size=112, height=128,
                          width=128, num_inference_steps=2, latents=inp["noise"].clone(), output_type="pt")
        out["edit_mini.npz"] = dict(feats=inp["feats"], noise=inp["noise"], image_latents=inp["image_latents"],
                                    src_image=inp["src_image"], latents_traj=torch.stack(traj), latents=lat,
                                    latents_no_image=lat0, latents_from_rgb=lat_img, image_pt=img)
    return out


ANYRES_BASE = 64
ANYRES_GRIDS = ['1x1', '1x2', '1x3', '2x1', '3x1', '1x4', '4x1', '2x2']       # eval_img2text_seed_x_i.py:57


def anyres_images():
    """Seeded uint8 test images (H, W): smooth gradients + noise so that resampling errors are visible."""
    rng = np.random.default_rng(77)
    out = []
    for H, W in ((100, 100), (90, 200), (260, 70), (130, 131), (37, 300)):
        yy, xx = np.mgrid[0:H, 0:W]
        base = np.stack([(xx * 255 // max(W - 1, 1)), (yy * 255 // max(H - 1, 1)), ((xx + yy) * 7 % 256)], -1)
        out.append(((base.astype(np.int32) + rng.integers(-40, 41, base.shape)).clip(0, 255)).astype(np.uint8))
    return out


def run_reference_preproc():
    """The reference's own process_anyres_image (src/inference/any_res.py:158-201) and get_transform
    (src/processer/transforms.py) executed on PIL images (torchvision.transforms restated in oracle/restated_preproc.py)."""
    refshim.install()
    sys.path.insert(0, os.path.join(refshim.REF_ROOT, "src", "inference"))
    from any_res import process_anyres_image
    from src.processer.transforms import get_transform
    from PIL import Image
    S = ANYRES_BASE
    grid_pinpoints = [[int(s.split('x')[0]) * S, int(s.split('x')[1]) * S] for s in ANYRES_GRIDS]
    tf = get_transform(type='clip', image_size=S, keep_ratio=False)
    arrs = {}
    for i, im in enumerate(anyres_images()):
        t, pos = process_anyres_image(Image.fromarray(im), tf, grid_pinpoints, S)
        arrs[f"img{i}"], arrs[f"out{i}"], arrs[f"pos{i}"] = im, t, pos
    im = anyres_images()[1]
    for name, kw in (("clip_keep", dict(type='clip', keep_ratio=True)), ("sd", dict(type='sd', keep_ratio=False)),
                     ("clipb_keep", dict(type='clipb', keep_ratio=True)), ("clipa", dict(type='clipa', keep_ratio=False))):
        arrs["tf_" + name] = get_transform(image_size=S, **kw)(Image.fromarray(im))
    return {"anyres_mini.npz": arrs}


# ---- path B orchestration: the reference's OWN ContinuousLVLM.generate, executed over oracle/hf_generate_shim.py -------
LVLM_VIT_DIM, LVLM_GRID, LVLM_HEADS = 128, 4, 2
LVLM_MIN_GAP = 0.08          # every free (non-forced) arg-max of a committed transcript wins by at least this much


def lvlm_cases(variant=0):
    """Inputs of the four generate() cases (seeded by `variant`; the fixture stores the inputs it was made with, the
    reference's OUTPUTS, and the lm_head rows that were re-aimed to steer the transcript). 16 resampled tokens per crop."""
    nq = LVLM_GRID * LVLM_GRID
    v = variant

    def crops(n, seed):
        return torch.randn(n, 36, LVLM_VIT_DIM, generator=torch.Generator().manual_seed(seed + 100 * v))
    cases = {}
    # (a) comprehension: 1 image = 2 crops (tile + global, any_res.py:185-189), ends on EOS
    ids = [1, 11, 12, 13 + v] + [0] * nq + [0] * nq + [21, 22, 23, 24, 25]
    m = torch.zeros(1, len(ids), dtype=torch.bool)
    m[0, 4:4 + 2 * nq] = True
    cases["comp2"] = dict(input_ids=[ids], ids_cmp_mask=m, image_embeds=crops(2, 5), embeds_cmp_mask=torch.tensor([True, True]),
                          patch_positions=torch.tensor([[0.0, 0.0], [0.5, 0.5]]), max_new_tokens=24, num_img_gen_tokens=16,
                          aim=(2, 9))                      # re-aim lm_head[EOS] so that new token #9 is EOS
    # (b) text → image: prompt STRING through the tokenizer (seed_x.py:151-152), no image input, default 64 image tokens,
    #     transcript = text, one image block, text until max_new_tokens
    cases["t2i"] = dict(prompt=f"31 32 33 34 35 36 37 38 39 40 41 {42 + v}", max_new_tokens=80, num_img_gen_tokens=64,
                        aim=(400, 4))
    # (c) any-res: 896-px-style input = 2x2 tiles + global = 5 crops, one of six embeds NOT selected (embeds_cmp_mask False:
    #     seed_x.py:173 indexes image_embeds_lm with it), input_ids as a LongTensor (seed_x.py:154-157), image block(s) out
    ids = [1, 51, 52 + v] + ([0] * nq + [53]) * 5 + [54, 55, 56]
    m = torch.zeros(1, len(ids), dtype=torch.bool)
    for c in range(5):
        m[0, 3 + c * (nq + 1):3 + c * (nq + 1) + nq] = True
    pp = torch.tensor([[0.0, 0.0], [0.0, 0.5], [0.7, 0.7], [0.5, 0.0], [0.5, 0.5], [0.5, 0.5]])
    cases["anyres5"] = dict(input_ids=torch.tensor([ids]), ids_cmp_mask=m, image_embeds=crops(6, 6),
                            embeds_cmp_mask=torch.tensor([True, True, False, True, True, True]), patch_positions=pp,
                            max_new_tokens=48, num_img_gen_tokens=16, aim=(400, 7))
    # (d) no image in, image block cut by max_new_tokens: <img> + part of the chain, no </img> → has_img_output False
    cases["truncated"] = dict(input_ids=[[1, 61, 62, 63, 64, 65, 66 + v]], max_new_tokens=12, num_img_gen_tokens=16, aim=(400, 5))
    return cases


def _lvlm_weights(ckpt16=False):
    """ckpt16: the weights a 16-bit (fp16) checkpoint holds — every tensor rounded to fp16-representable values (the `_ckpt16` fixtures)."""
    cfg = weights.MINI_LLM
    sd_llm, sd_agent = weights.llama_sd(cfg), weights.agent_sd(cfg, LVLM_VIT_DIM, in_grid=LVLM_GRID, out_grid=LVLM_GRID)
    if ckpt16:
        sd_llm = {k: v.to(torch.float16).float() for k, v in sd_llm.items()}
        sd_agent = {k: v.to(torch.float16).float() for k, v in sd_agent.items()}
    return cfg, sd_llm, sd_agent


def _run_ref_generate(sd_llm, sd_agent, case, ckpt16=False):
    """One call of the reference's ContinuousLVLM.generate (fp32, CPU). Returns its result dict + what llm.generate produced.
    ckpt16: the RoPE tables of the reference's fp16 runs are put into its rotary buffers (see run_reference_llama_ckpt16)."""
    from oracle import hf_generate_shim as hs
    cfg = weights.MINI_LLM
    m = hs.build_reference_lvlm(cfg, sd_llm, sd_agent, LVLM_VIT_DIM, LVLM_GRID, LVLM_GRID, LVLM_HEADS)
    if ckpt16:
        with torch.no_grad():
            for layer in m.llm.model.layers:
                re = layer.self_attn.rotary_emb
                re.cos_cached.copy_(re.cos_cached.to(torch.float16).float())
                re.sin_cached.copy_(re.sin_cached.to(torch.float16).float())
    rec = {}
    inner = m.llm.generate

    def recording_generate(**kw):                         # observes (does not alter) the call made at seed_x.py:184-189
        rec["kwargs"] = {k: v for k, v in kw.items() if k not in ("input_ids", "inputs_embeds", "logits_processor")}
        rec["out"] = inner(output_scores=True, **kw)
        return rec["out"]
    m.llm.generate = recording_generate
    kw = {k: v for k, v in case.items() if k != "aim"}
    with torch.no_grad():
        out = m.generate(hs.StubTokenizer(), dtype=torch.float32, device="cpu", **kw)
    return out, rec


def _aim_row(sd_llm, sd_agent, case, ckpt16=False):
    """Re-aims ONE lm_head row so that the seeded random-weight model emits EOS / <img> as new token #step and not before:
    the minimum-norm row w with w·h_t = -4 for the final states h_t of the earlier steps and w·h_step = (that step's best
    logit) + 3, h_t taken from the reference's own run. Other rows are untouched, so the transcript before #step is too."""
    row, step = case["aim"]
    _, rec = _run_ref_generate(sd_llm, sd_agent, dict(case, max_new_tokens=step + 1), ckpt16)
    hs_ = torch.stack([rec["out"].hidden_states[t][-1][0, -1] for t in range(step + 1)])       # [step+1, H]
    tgt = torch.full((step + 1,), -4.0)
    tgt[step] = float(rec["out"].scores[step][0].max()) + 3.0
    w = torch.linalg.pinv(hs_.double()) @ tgt.double()
    sd_llm["lm_head.weight"][row] = w.float().to(torch.float16).float() if ckpt16 else w.float()    # (a checkpoint row is 16-bit too)
    return row, sd_llm["lm_head.weight"][row].clone()


def run_reference_generate(max_variants=40, ckpt16=False):
    """{file: arrays}. Per case the input variant is advanced until every free arg-max of the reference's transcript wins by
    ≥ LVLM_MIN_GAP (so a 16-bit implementation is expected to reproduce the ids exactly); the chosen inputs travel in the fixture."""
    from oracle import hf_generate_shim as hs
    cfg, sd_llm0, sd_agent = _lvlm_weights(ckpt16)
    tok = hs.StubTokenizer()
    arrs = {}
    for name in lvlm_cases():
        for variant in range(max_variants):
            case = lvlm_cases(variant)[name]
            sd_llm = {k: v.clone() for k, v in sd_llm0.items()}
            rows = dict([_aim_row(sd_llm, sd_agent, case, ckpt16)]) if case["aim"] is not None else {}
            out, rec = _run_ref_generate(sd_llm, sd_agent, case, ckpt16)
            o = rec["out"]
            top2 = torch.stack([torch.topk(s[0], 2).values for s in o.scores])
            gap, std = top2[:, 0] - top2[:, 1], torch.stack([s[0].std() for s in o.scores])
            free = gap < 9.0                                                # forced steps win by exactly 10 (generation.py:26)
            if free.any() and float(gap[free].min()) >= LVLM_MIN_GAP:
                break
        else:
            raise RuntimeError(f"{name}: no variant with a clear transcript")
        assert rec["kwargs"]["do_sample"] is False and rec["kwargs"]["output_hidden_states"] and rec["kwargs"]["return_dict_in_generate"]
        if "prompt" in case:
            in_ids = torch.tensor([tok.encode(case["prompt"], add_special_tokens=True)])
        else:
            in_ids = torch.as_tensor(case["input_ids"]).reshape(1, -1)
        n_in = in_ids.shape[1]
        seq = o.sequences[0]
        assert torch.equal(seq[:n_in], in_ids[0])                          # quirk 14: the prompt ids lead .sequences
        last_hidden = torch.cat([hs_[-1] for hs_ in o.hidden_states], dim=1)[0, n_in:, :]   # seed_x.py:196-197
        a = dict(variant=np.array(variant), input_ids=in_ids, max_new_tokens=np.array(case["max_new_tokens"]),
                 num_img_gen_tokens=np.array(case["num_img_gen_tokens"]), prompt=np.array(case.get("prompt", "")),
                 sequences=seq, generate_ids=seq[n_in:], top2_gap=gap, score_std=std, last_hidden_states=last_hidden,
                 text=np.array(out["text"]), has_img_output=np.array(out["has_img_output"]),
                 num_gen_imgs=np.array(out["num_gen_imgs"]),
                 img_gen_feat=out["img_gen_feat"] if out["img_gen_feat"] is not None else torch.zeros(0))
        for k in ("ids_cmp_mask", "image_embeds", "embeds_cmp_mask", "patch_positions"):
            if k in case:
                a[k] = case[k]
        for r, v in rows.items():
            a[f"lm_head_row_{r}"] = v
        print(f"  {name}: variant {variant}, {len(seq) - n_in} new tokens {seq[n_in:].tolist()}, min free gap {float(gap[free].min()):.3f}, "
              f"text {out['text']!r}, images {out['num_gen_imgs']}")
        arrs.update({f"{name}.{k}": v for k, v in a.items()})
    return {("lvlm_generate_mini_ckpt16.npz" if ckpt16 else "lvlm_generate_mini.npz"): arrs}


LVLM_CASE_NAMES = ("comp2", "t2i", "anyres5", "truncated")


def lvlm_case_weights(name, gold, ckpt16=False):
    """(cfg, sd_llm, sd_agent) of a committed case: the seeded weights + the fixture's re-aimed lm_head rows."""
    cfg, sd_llm, sd_agent = _lvlm_weights(ckpt16)
    for k in gold.files:
        if k.startswith(name + ".lm_head_row_"):
            sd_llm["lm_head.weight"][int(k.rsplit("_", 1)[1])] = torch.as_tensor(np.asarray(gold[k]))
    return cfg, sd_llm, sd_agent


def lvlm_case_inputs(name, gold):
    """generate() keyword arguments of a committed case, rebuilt from the fixture."""
    kw = dict(max_new_tokens=int(gold[f"{name}.max_new_tokens"]), num_img_gen_tokens=int(gold[f"{name}.num_img_gen_tokens"]))
    if str(gold[f"{name}.prompt"]):
        kw["prompt"] = str(gold[f"{name}.prompt"])
    else:
        kw["input_ids"] = torch.as_tensor(gold[f"{name}.input_ids"])
    for k in ("ids_cmp_mask", "image_embeds", "embeds_cmp_mask", "patch_positions"):
        if f"{name}.{k}" in gold.files:
            kw[k] = torch.as_tensor(gold[f"{name}.{k}"])
    return kw


LLAMA16_TOKS = [17, 44, 301]


def run_reference_llama_ckpt16():
    """The reference's LlamaForCausalLM on the weights a 16-bit checkpoint holds and with the RoPE tables it uses when its scripts run it in
    16 bit — executed in fp32 on the CPU: the ground truth of the decoder's precise mode at north_star's 1e-3 (tests/golden/
    llama_mini_ckpt16.npz). The reference casts cos / sin to the activation dtype on every call (modeling_llama_xformer.py:128-131); here the
    module runs in fp32, so the rounded tables are put into its `cos_cached` / `sin_cached` buffers (values, not code: what an fp16 / bf16
    run of the same module multiplies with)."""
    from transformers import LlamaConfig
    mods = refshim.reference_modules()
    cfg = weights.MINI_LLM
    out = {}
    x = torch.randn(1, 21, cfg["hidden_size"], generator=torch.Generator().manual_seed(3030)) * 0.5
    out["x"] = x
    out["toks"] = np.array(LLAMA16_TOKS)
    with torch.no_grad():
        for tag, dt in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
            sd = {k: v.to(dt).float() for k, v in weights.llama_sd(cfg).items()}
            llm = mods["LlamaForCausalLM"](LlamaConfig(**cfg)).eval()
            full = dict(llm.state_dict())
            full.update(sd)
            llm.load_state_dict(full, strict=True)
            for layer in llm.model.layers:
                re = layer.self_attn.rotary_emb
                re.cos_cached.copy_(re.cos_cached.to(dt).float())
                re.sin_cached.copy_(re.sin_cached.to(dt).float())
            T = x.shape[1]
            o = llm(inputs_embeds=x, attention_mask=torch.ones(1, T, dtype=torch.long), use_cache=True, output_hidden_states=True,
                    return_dict=True)
            out[tag + ".logits"], out[tag + ".hidden"] = o.logits, o.hidden_states[-1]
            pkv, steps = o.past_key_values, []
            for i, t in enumerate(LLAMA16_TOKS):
                o = llm(input_ids=torch.tensor([[t]]), attention_mask=torch.ones(1, T + 1 + i, dtype=torch.long), past_key_values=pkv,
                        use_cache=True, return_dict=True)
                pkv = o.past_key_values
                steps.append(o.logits[0, -1])
            out[tag + ".step_logits"] = torch.stack(steps)
    return {"llama_mini_ckpt16.npz": out}


def run_reference_modules_ckpt16():
    """ViT (both head dims), the LLM-side Resampler and ResamplerXLV2 of the reference, executed in fp32 on the weights a 16-bit (fp16)
    checkpoint holds (tests/golden/modules_mini_ckpt16.npz): the same modules as vit_hd*.npz / resampler_mini.npz / xlv2_mini.npz, without the
    0.8-1.2e-3 that rounding fp32 fixture weights to fp16 costs — so that the HIP modules are compared with the REFERENCE's numbers at
    north_star's tolerance."""
    mods = refshim.reference_modules()
    r16 = lambda sd: {k: v.to(torch.float16).float() for k, v in sd.items()}
    g = torch.Generator().manual_seed(4040)
    out = {}
    with torch.no_grad():
        for tag, cfg in (("vit_hd128", weights.MINI_VIT), ("vit_hd104", weights.MINI_VIT_104)):
            m = mods["VisionTransformerWithAttnPool"](**cfg).eval()
            m.load_state_dict(r16(weights.vit_sd(cfg)), strict=True)
            x = torch.randn(2, 3, cfg["image_size"], cfg["image_size"], generator=g)
            out[tag + ".x"], out[tag + ".y"] = x, m(x).float()
        r = mods["Resampler"](grid_size=4, embed_dim=320, num_heads=2, kv_dim=256).eval()
        r.load_state_dict(r16(weights.resampler_sd(weights._g(7), "", 4, 320, 256)), strict=True)
        x = torch.randn(2, 16, 256, generator=g)
        out["resampler.x"], out["resampler.y"] = x, r(x)
        cfgx = weights.MINI_XLV2
        m = mods["ResamplerXLV2"](normalize=False, **cfgx).eval()
        m.load_state_dict(r16(weights.xlv2_sd(cfgx, pre="")), strict=True)
        x = torch.randn(2, 24, cfgx["embedding_dim"], generator=g)
        out["xlv2.x"] = x
        out["xlv2.prompt"], out["xlv2.pooled"] = m(x)
    return {"modules_mini_ckpt16.npz": out}


def main_llama16():
    for name, arrs in run_reference_llama_ckpt16().items():
        _save(name, **arrs)
    for name, arrs in run_reference_modules_ckpt16().items():
        _save(name, **arrs)
    for name, arrs in run_reference_generate(ckpt16=True).items():
        _save(name, **arrs)


def main_generate():
    for name, arrs in run_reference_generate().items():
        _save(name, **arrs)


def main_detok():
    for name, arrs in run_reference_preproc().items():
        _save(name, **arrs)
    for name, arrs in run_reference_detok().items():
        _save(name, **arrs)


if __name__ == "__main__":
    if not refshim.available():
        raise SystemExit("/root/reference is not available: golden fixtures can only be generated in the build container")
    if "--generate-only" in sys.argv:
        main_generate()
        raise SystemExit(0)
    if "--llama16-only" in sys.argv:
        main_llama16()
        raise SystemExit(0)
    if "--detok-only" not in sys.argv:
        main()
        main_generate()
        main_llama16()
    main_detok()
