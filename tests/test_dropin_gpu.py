"""Drop-in boundary on the GPU: every component is loaded FROM DISK in the reference's checkpoint
layouts through `_target_` strings, the reference's script statements are replayed unchanged (PIL image in →
`generated_images[0].save(...)` out) and the result is checked against the CPU oracle."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

from oracle import restated_adapter as ra, restated_preproc as rp, restated_unet as ru, restated_vae as rv, weights
from tests._pretrained_tree import StubTokenizer, write_tree
from tests.test_dropin_cpu import replay_detokenizer_script

pytestmark = pytest.mark.gpu


def _src_image():
    rng = np.random.default_rng(9)
    yy, xx = np.mgrid[0:150, 0:200]
    a = np.stack([xx * 255 // 199, yy * 255 // 149, (xx + yy) % 256], -1) + rng.integers(-30, 31, (150, 200, 3))
    return Image.fromarray(a.clip(0, 255).astype(np.uint8))


def test_eval_seed_x_detokenizer_flow_from_disk(dev, tmp_path):
    """eval_seed_x_detokenizer.py:22-61: image → ViT → ResamplerXLV2 → CFG-2 Euler loop → VAE → PIL, saved to disk.
    Oracle: Pillow/torchvision transform → restated adapter + UNet loop → restated VAE decode → uint8."""
    paths = write_tree(tmp_path)
    img = _src_image()
    noise = torch.randn(1, 4, 16, 16, generator=torch.Generator().manual_seed(3))
    adapter, images = replay_detokenizer_script(paths, "cuda", torch.float16, img, 3, height=128, width=128,
                                                latents=noise.clone())
    assert isinstance(images, list) and isinstance(images[0], Image.Image) and images[0].size == (128, 128)
    out = str(tmp_path / "recon.jpg")
    images[0].save(out)                                                   # the script's last statement (:61)
    assert os.path.getsize(out) > 0
    V, X, A = weights.DETOK_VIT, weights.DETOK_XLV2, weights.DETOK_VAE
    x = rp.clip_transform(V["image_size"])(img)[None]
    u4 = weights.detok_unet_cfg(4)
    lat = ra.adapter_generate(weights.vit_sd(V), V, weights.xlv2_sd(X), X, ru.unet_sd(u4), u4, noise, 3, image_tensor=x,
                              height=128, width=128)
    ref = rp.postprocess_pil(rv.vae_decode(rv.vae_sd(A), A, lat / A["scaling_factor"]))[0]
    d = np.abs(np.asarray(images[0], dtype=np.int32) - np.asarray(ref, dtype=np.int32))
    print(f"detokenizer script flow: mean |Δ| {d.mean():.3f} / 255, max {d.max()}")
    assert d.mean() < 1.0 and d.max() <= 12
    lat_hip = adapter.generate(img, num_inference_steps=3, height=128, width=128, latents=noise.clone(), output_type="latent")
    e = ((lat_hip.float().cpu() - lat).norm() / lat.norm()).item()
    assert e < 5e-3, e


def test_eval_text2img_and_edit_flows_from_disk(dev, tmp_path):
    """eval_text2img_seed_x_i.py:36-93 (LLM → image features → t2i adapter → PIL) and the adapter call of
    eval_img2edit_seed_x_edit.py:146-149 (PIL `latent_image`), components instantiated from the overlay YAMLs."""
    from seedx_amd import dropin
    from seedx_amd import image_ops
    paths = write_tree(tmp_path)
    device, dtype = "cuda", torch.float16
    tokenizer = StubTokenizer()
    image_transform = dropin.instantiate(dropin.load_config(paths["image_transform_cfg_path"]))
    visual_encoder = dropin.instantiate(dropin.load_config(paths["visual_encoder_cfg_path"]))
    visual_encoder.eval().to(device, dtype=dtype)
    llm = dropin.instantiate(dropin.load_config(paths["llm_cfg_path"]), torch_dtype=dtype)
    agent_model = dropin.instantiate(dropin.load_config(paths["agent_cfg_path"]), llm=llm)
    agent_model.eval().to(device, dtype=dtype)
    prompt_ids = tokenizer.encode("[INST] Generate an image: a red cube [/INST]\n", add_special_tokens=False)
    input_ids = torch.tensor([tokenizer.bos_token_id] + prompt_ids).to(device, dtype=torch.long).unsqueeze(0)
    output = agent_model.generate(tokenizer=tokenizer, input_ids=input_ids, num_img_gen_tokens=16, max_new_tokens=40,
                                  eos_token_id=None, force_image_at=2)
    assert output["has_img_output"] and output["img_gen_feat"].shape == (1, 16, 256)
    # t2i adapter from the first-stage checkpoint
    adapter, _ = replay_detokenizer_script(paths, device, dtype, _src_image(), 1, height=128, width=128)
    images = adapter.generate(image_embeds=output["img_gen_feat"].to(device), num_inference_steps=3, height=128, width=128,
                              input_image_size=112)
    images[0].save(str(tmp_path / "t2i.png"))
    assert images[0].size == (128, 128)
    # edit adapter from the second-stage checkpoint, PIL source image (eval_img2edit_seed_x_edit.py:101,147)
    src = _src_image().resize((128, 128))
    edit, images = replay_detokenizer_script(paths, device, dtype, None, 2, edit=True, height=128, width=128, seed=42,
                                             image_embeds=output["img_gen_feat"], latent_image=src, input_image_size=112)
    images[0].save(str(tmp_path / "edit.jpg"))
    # parity of that edit call: same noise through the oracle (seeded noise is drawn on the GPU: fetch it back)
    noise = edit._noise(42, 128, 128, 1).cpu()
    lat = edit.generate(image_embeds=output["img_gen_feat"], latent_image=src, num_inference_steps=2, height=128, width=128,
                        input_image_size=112, latents=noise.clone(), output_type="latent")
    V, X, A = weights.DETOK_VIT, weights.DETOK_XLV2, weights.DETOK_VAE
    u8c = weights.detok_unet_cfg(8)
    il = ra.edit_image_latents(rv.vae_encoder_sd(A), A, src)
    ref = ra.adapter_generate(weights.vit_sd(V), V, weights.xlv2_sd(X), X, ru.unet_sd(u8c), u8c, noise, 2,
                              image_embeds=output["img_gen_feat"].float().cpu(), image_latents=il, height=128, width=128)
    e = ((lat.float().cpu() - ref).norm() / ref.norm()).item()
    print(f"edit script flow (PIL source) rel-L2 {e:.3e}")
    assert e < 6e-3
    # marker mask + any-res of eval_img2text_seed_x_i.py:131-160 on the device
    grid = [[112, 112], [112, 224], [224, 112], [224, 224]]
    image_tensor, patch_pos = image_ops.process_anyres_image(_src_image(), image_transform, grid, 112)
    assert image_tensor.is_cuda and image_tensor.shape[1:] == (3, 112, 112) and patch_pos.shape[0] == image_tensor.shape[0]
    feats = visual_encoder(image_tensor)
    assert feats.shape == (image_tensor.shape[0], 64, 256)
