"""CPU restatement of the SDXLAdapter front ends (TEST INFRASTRUCTURE; composes oracle/restated*.py).

Follows src/models/detokenizer/adapter_modules.py:
  * get_image_embeds (:96-130): three mutually exclusive inputs; pooling ONLY on the image_embeds branch (:109-116)
  * SDXLAdapter.forward (:39-52): resampler → one UNet forward → MSE against the noise
  * SDXLAdapter.generate (:132-169) → t2i loop, order [uncond, text]
  * SDXLAdapterWithLatentImage.generate (:249-287) → edit loop (pipeline_stable_diffusion_xl_t2i_edit.py:900-963)
"""
import torch
import torch.nn.functional as F

from . import restated, restated_unet as ru


def get_image_embeds(sd_vit, cfg_vit, sd_x, cfg_x, image_tensor=None, image_embeds=None, vit_down=True, image_size=None):
    assert (image_tensor is None) != (image_embeds is None)
    if image_tensor is not None:
        x = torch.cat([image_tensor, torch.zeros_like(image_tensor)], dim=0)          # :103-106
        feats = restated.vit_forward(sd_vit, cfg_vit, x)                               # no pooling on this branch (:108)
    else:
        s = image_size or cfg_vit["image_size"]
        neg = restated.vit_forward(sd_vit, cfg_vit, torch.zeros(1, 3, s, s, device=image_embeds.device))   # :110-111
        if vit_down:
            neg = F.avg_pool1d(neg.permute(0, 2, 1), kernel_size=4, stride=4).permute(0, 2, 1)   # :112-115
        feats = torch.cat([image_embeds.float(), neg.expand(image_embeds.shape[0], -1, -1)], dim=0)   # :116
    prompt, pooled = restated.resampler_xlv2_forward(sd_x, cfg_x, feats)               # :118-120 (discrete model = identity)
    n = prompt.shape[0] // 2
    return prompt[:n], prompt[n:], pooled[:n], pooled[n:]                              # :122-124


def adapter_forward(sd_x, cfg_x, sd_unet, cfg_unet, noisy_latents, timesteps, image_embeds, noise, time_ids):
    """adapter_modules.py:39-52: (`text_embeds` is ignored by the reference) → (total_loss, noise_pred)."""
    prompt, pooled = restated.resampler_xlv2_forward(sd_x, cfg_x, image_embeds.float())        # :41
    noise_pred = ru.unet_forward(sd_unet, cfg_unet, noisy_latents.float(), timesteps, prompt, pooled, time_ids)   # :43-45
    return F.mse_loss(noise_pred.float(), noise.float(), reduction="mean"), noise_pred     # :48


def adapter_generate(sd_vit, cfg_vit, sd_x, cfg_x, sd_unet, cfg_unet, latents, steps, image_tensor=None,
                     image_embeds=None, vit_down=True, guidance_scale=7.5, image_latents=None,
                     image_guidance_scale=1.5, height=1024, width=1024):
    """latents: raw N(0,1) noise [1,4,h,w] (scaled by init_noise_sigma here, as the pipelines do). image_latents given
    → edit variant. Returns final latents."""
    pe, pe_neg, pool, pool_neg = get_image_embeds(sd_vit, cfg_vit, sd_x, cfg_x, image_tensor, image_embeds, vit_down)
    _, _, init = ru.euler_tables(steps)
    tid = torch.tensor([[height, width, 0, 0, height, width]], dtype=torch.float32, device=latents.device)   # pipeline…:554-566
    fn = lambda s, t, e, p, ti: ru.unet_forward(sd_unet, cfg_unet, s, t, e, p, ti)
    lat0 = latents.float() * init
    if image_latents is None:
        return ru.t2i_loop(fn, lat0, pe, pe_neg, pool, pool_neg, tid, steps, guidance_scale)
    il3 = torch.cat([image_latents, image_latents, torch.zeros_like(image_latents)], dim=0)   # :544-546
    return ru.edit_loop(fn, lat0, il3, pe, pe_neg, pool, pool_neg, tid, steps, guidance_scale, image_guidance_scale)


def vae_preprocess(image):
    """VaeImageProcessor.preprocess [ext] as called at pipeline…:823: 4-channel tensors are latents and pass through;
    RGB in [0,1] is mapped to [-1,1] (already-negative input is left alone); PIL → float/255 first (no resize here:
    callers hand multiples of 8)."""
    import numpy as np
    if not torch.is_tensor(image):
        image = torch.from_numpy(np.asarray(image.convert("RGB"), dtype=np.float32) / 255.0).permute(2, 0, 1)[None]
    if image.ndim == 3:
        image = image[None]
    if image.shape[1] == 4:
        return image
    return 2.0 * image - 1.0 if image.min() >= 0 else image


def edit_image_latents(sd_vae_enc, cfg_vae, latent_image):
    """prepare_image_latents (pipeline…:488-552): latents pass through; RGB is encoded with the fp32 VAE and the
    distribution's mode() is used UN-scaled (:523, no scaling_factor)."""
    from . import restated_vae as rv
    x = vae_preprocess(latent_image)
    return x if x.shape[1] == 4 else rv.vae_encode_mode(sd_vae_enc, cfg_vae, x)


def decode_to_pt(sd_vae_dec, cfg_vae, latents):
    """pipeline…:965-986 with output_type="pt": decode(latents / scaling_factor) → (x/2+0.5).clamp(0,1)."""
    from . import restated_vae as rv
    return (rv.vae_decode(sd_vae_dec, cfg_vae, latents / cfg_vae["scaling_factor"]) / 2 + 0.5).clamp(0, 1)
