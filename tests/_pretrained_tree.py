"""Writes a miniature `pretrained/` + `configs/` tree in the REFERENCE's on-disk layouts with the seeded
oracle weights, so that the reference's script flows can be replayed end to end: Qwen ViT .pt, HF Llama directory,
agent .bin, diffusers unet/ vae/ scheduler/ directories, first- and second-stage de-tokenizer .bin, and overlay YAMLs
(the shipped configs/*.yaml with mini dimensions and paths pointing into the tree). TEST INFRASTRUCTURE."""
import json
import os

import torch
import yaml

from oracle import restated_unet as ru, restated_vae as rv, weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VIT_DIM = 256                                            # ViT output_dim = resampler kv_dim = XLV2 embedding_dim
IMG_IDS = list(range(400, 466))                          # <img>, <img_00000>..<img_00063>, </img> of the stub vocabulary


class StubTokenizer:
    """encode()/decode()/bos/eos of the LlamaTokenizer the scripts instantiate (host-side sentencepiece stays the reference's)."""
    bos_token_id, eos_token_id, pad_token_id = 1, 2, 0

    def encode(self, s, add_special_tokens=False):
        import re
        out = []
        for tok in re.findall(r"<img_\d{5}>|<img>|</img>|<patch>|</patch>|[^<]+|<", s):
            if tok == "<img>":
                out.append(IMG_IDS[0])
            elif tok == "</img>":
                out.append(IMG_IDS[-1])
            elif tok == "<patch>":
                out.append(398)
            elif tok == "</patch>":
                out.append(399)
            elif tok.startswith("<img_"):
                out.append(IMG_IDS[1] + int(tok[5:10]))
            else:
                out.extend(3 + (ord(c) % 300) for c in tok[:24])
        return out

    def batch_decode(self, ids, skip_special_tokens=False):
        return [" ".join(str(int(i)) for i in row) for row in ids]

    def decode(self, ids, skip_special_tokens=False):
        return " ".join(str(int(i)) for i in ids)


def write_tree(root, vit_cfg=None):
    """Returns a dict of the overlay-config paths (keys = the script variable names) and component configs."""
    from safetensors.torch import save_file
    root = str(root)
    P = os.path.join(root, "pretrained")
    vit_cfg = dict(vit_cfg or weights.DETOK_VIT)
    assert vit_cfg["output_dim"] == VIT_DIM
    os.makedirs(os.path.join(P, "QwenViT"), exist_ok=True)
    torch.save(weights.vit_sd(vit_cfg), os.path.join(P, "QwenViT", "qwen_vit_G.pt"))
    # HF Llama dir
    lcfg = weights.MINI_LLM
    ldir = os.path.join(P, "seed_x_i", "llm")
    os.makedirs(ldir, exist_ok=True)
    json.dump(dict(lcfg, architectures=["LlamaForCausalLM"], model_type="llama"), open(os.path.join(ldir, "config.json"), "w"))
    save_file({k: v.contiguous() for k, v in weights.llama_sd(lcfg).items()}, os.path.join(ldir, "model-00001-of-00001.safetensors"))
    os.makedirs(os.path.join(P, "seed_x_i", "agent"), exist_ok=True)
    torch.save(weights.agent_sd(lcfg, VIT_DIM, in_grid=4, out_grid=4), os.path.join(P, "seed_x_i", "agent", "pytorch_model.bin"))
    # SDXL base: unet / vae / scheduler
    ucfg = weights.detok_unet_cfg(4)
    sdxl = os.path.join(P, "stable-diffusion-xl-base-1.0")
    for sub in ("unet", "vae", "scheduler"):
        os.makedirs(os.path.join(sdxl, sub), exist_ok=True)
    boc = ucfg["block_out_channels"]
    json.dump(dict(in_channels=4, out_channels=4, block_out_channels=list(boc), layers_per_block=ucfg["layers_per_block"],
                   transformer_layers_per_block=list(ucfg["transformer_layers"]), attention_head_dim=list(ucfg["heads"]),
                   cross_attention_dim=ucfg["cross_attention_dim"], addition_time_embed_dim=ucfg["addition_time_embed_dim"],
                   projection_class_embeddings_input_dim=ucfg["pooled_dim"] + 6 * ucfg["addition_time_embed_dim"],
                   norm_num_groups=ucfg["norm_groups"], sample_size=16,
                   down_block_types=["DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"],
                   up_block_types=["CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"]),
              open(os.path.join(sdxl, "unet", "config.json"), "w"))
    base_sd = ru.unet_sd(ucfg, seed=99)                                     # "stock SDXL": differs from the adapter's UNet
    save_file({k: v.contiguous() for k, v in base_sd.items()}, os.path.join(sdxl, "unet", "diffusion_pytorch_model.safetensors"))
    A = weights.DETOK_VAE
    json.dump(dict(in_channels=3, out_channels=3, block_out_channels=list(A["block_out_channels"]),
                   layers_per_block=A["layers_per_block"], latent_channels=4, norm_num_groups=32, scaling_factor=0.13025,
                   force_upcast=True, act_fn="silu"), open(os.path.join(sdxl, "vae", "config.json"), "w"))
    save_file({k: v.contiguous() for k, v in dict(rv.vae_sd(A), **rv.vae_encoder_sd(A)).items()},
              os.path.join(sdxl, "vae", "diffusion_pytorch_model.safetensors"))
    json.dump(dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                   steps_offset=1, timestep_spacing="leading", prediction_type="epsilon"),
              open(os.path.join(sdxl, "scheduler", "scheduler_config.json"), "w"))
    # de-tokenizer checkpoints: first stage = resampler + FULL unet; second stage = resampler + 8-channel unet
    X = weights.DETOK_XLV2
    for stage, in_ch in (("first_stage", 4), ("second_stage", 8)):
        d = os.path.join(P, "seed_detokenizer", stage)
        os.makedirs(d, exist_ok=True)
        ck = dict(weights.xlv2_sd(X, pre="resampler."))
        ck.update({"unet." + k: v for k, v in ru.unet_sd(weights.detok_unet_cfg(in_ch)).items()})
        torch.save(ck, os.path.join(d, "pytorch_model.bin"))
    # overlay YAMLs = the shipped configs with mini dimensions + paths into this tree
    C = os.path.join(root, "configs")

    def overlay(rel, **patch):
        cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", rel)))
        for k, v in patch.items():
            if isinstance(v, dict) and isinstance(cfg.get(k), dict):
                cfg[k].update(v)
            else:
                cfg[k] = v
        for k in ("pretrained_model_path", "pretrained_model_name_or_path"):
            if k in cfg and "tokenizer" not in rel:
                cfg[k] = os.path.join(root, cfg[k])
        out = os.path.join(C, rel)
        os.makedirs(os.path.dirname(out), exist_ok=True)
        yaml.safe_dump(cfg, open(out, "w"))
        return out
    H = lcfg["hidden_size"]
    xl = {k: X[k] for k in ("dim", "depth", "dim_head", "heads", "num_queries", "embedding_dim", "output1_dim", "output2_dim", "ff_mult")}
    paths = dict(
        visual_encoder_cfg_path=overlay("visual_encoder/qwen_vitg_448.yaml", **{k: vit_cfg[k] for k in
                                        ("image_size", "patch_size", "width", "layers", "heads", "mlp_ratio", "output_dim")},
                                        n_queries=vit_cfg["n_queries"]),
        image_transform_cfg_path=overlay("processer/qwen_448_transform.yaml", image_size=vit_cfg["image_size"]),
        llm_cfg_path=overlay("clm_models/llm_seed_x_i.yaml"),
        agent_cfg_path=overlay("clm_models/agent_seed_x_i.yaml",
                               input_resampler=dict(grid_size=4, embed_dim=H, num_heads=2, kv_dim=VIT_DIM),
                               output_resampler=dict(grid_size=4, embed_dim=VIT_DIM, num_heads=2, kv_dim=H)),
        adapter_cfg_path=overlay("sdxl_adapter/sdxl_qwen_vit_resampler_l4_q64_pretrain_no_normalize.yaml", resampler=xl),
        edit_adapter_cfg_path=overlay("sdxl_adapter/sdxl_qwen_vit_resampler_l4_q64_full_with_latent_image_pretrain_no_normalize.yaml",
                                      resampler=xl),
        discrete_model_cfg_path=overlay("discrete_model/discrete_identity.yaml"),
        diffusion_model_path=sdxl)
    return paths
