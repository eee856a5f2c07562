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


def main_detok():
    for name, arrs in run_reference_preproc().items():
        _save(name, **arrs)
    for name, arrs in run_reference_detok().items():
        _save(name, **arrs)


if __name__ == "__main__":
    if not refshim.available():
        raise SystemExit("/root/reference is not available: golden fixtures can only be generated in the build container")
    if "--detok-only" not in sys.argv:
        main()
    main_detok()
