"""Seeded random weights with the REFERENCE's state-dict key names (test infrastructure).

No checkpoints exist in this environment, so parity is measured with seeded random
initialisation at "mini" dimensions (seconds on CPU) and at the real dimensions (GPU-side generation for bench).
Scales are chosen so activations stay O(1) through the depth of each network; LayerNorm/GroupNorm affine
parameters are non-trivial so that affine bugs are visible.
"""
import math

import torch

MINI_VIT = dict(image_size=112, patch_size=14, width=256, layers=3, heads=2, mlp_ratio=2.0, n_queries=16,
                output_dim=256)  # head_dim 128 → pool heads = 2; ViT head_dim = 128
MINI_VIT_104 = dict(image_size=112, patch_size=14, width=832, layers=2, heads=8, mlp_ratio=4.9231, n_queries=16,
                    output_dim=256)  # head_dim 104 like ViT-G; mlp = int(832*4.9231) = 4096
FULL_VIT = dict(image_size=448, patch_size=14, width=1664, layers=48, heads=16, mlp_ratio=4.9231, n_queries=256,
                output_dim=4096)  # configs/visual_encoder/qwen_vitg_448.yaml

MINI_LLM = dict(hidden_size=256, intermediate_size=704, num_hidden_layers=3, num_attention_heads=2, vocab_size=500,
                rms_norm_eps=1e-5, max_position_embeddings=512)
FULL_LLM = dict(hidden_size=5120, intermediate_size=13824, num_hidden_layers=40, num_attention_heads=40,
                vocab_size=32330, rms_norm_eps=1e-5, max_position_embeddings=4096)  # Llama-2-13B dims, SURVEY §8a-B

MINI_XLV2 = dict(dim=128, depth=2, dim_head=64, heads=2, num_queries=16, embedding_dim=256, output1_dim=64,
                 output2_dim=128, ff_mult=4)
FULL_XLV2 = dict(dim=1024, depth=4, dim_head=64, heads=16, num_queries=64, embedding_dim=4096, output1_dim=768,
                 output2_dim=1280, ff_mult=4)  # configs/sdxl_adapter/*.yaml

# mini de-tokenizer stack of tests/golden/{t2i,edit}_mini.npz (oracle/gen_golden.py runs the reference adapters on it)
DETOK_VIT = dict(image_size=112, patch_size=14, width=256, layers=2, heads=2, mlp_ratio=2.0, n_queries=64, output_dim=256)
DETOK_XLV2 = dict(MINI_XLV2, embedding_dim=256)
DETOK_VAE = dict(latent_channels=4, out_channels=3, block_out_channels=(64, 64, 128, 128), layers_per_block=1,
                 norm_groups=32, scaling_factor=0.13025)


def detok_unet_cfg(in_ch):
    from .restated_unet import MINI_UNET
    return dict(MINI_UNET, in_channels=in_ch, cross_attention_dim=192, pooled_dim=128)


def _g(seed):
    return torch.Generator(device="cpu").manual_seed(seed)


def _lin(g, out_f, in_f, gain=1.0):
    return torch.randn(out_f, in_f, generator=g) * (gain / math.sqrt(in_f))


def _vec(g, n, std=0.02):
    return torch.randn(n, generator=g) * std


def _norm(g, sd, name, n):
    sd[name + ".weight"] = 1.0 + 0.1 * torch.randn(n, generator=g)
    sd[name + ".bias"] = 0.05 * torch.randn(n, generator=g)


def resampler_sd(g, pre, grid, embed_dim, kv_dim, sd=None):
    """Resampler parameters (qwen_visual.py:102-125)."""
    from .restated import sincos_2d
    sd = {} if sd is None else sd
    sd[pre + "pos_embed"] = sincos_2d(embed_dim, grid)
    sd[pre + "query"] = torch.randn(grid * grid, embed_dim, generator=g) * 0.5
    if kv_dim != embed_dim:
        sd[pre + "kv_proj.weight"] = _lin(g, embed_dim, kv_dim)
    sd[pre + "attn.in_proj_weight"] = _lin(g, 3 * embed_dim, embed_dim)
    sd[pre + "attn.in_proj_bias"] = _vec(g, 3 * embed_dim)
    sd[pre + "attn.out_proj.weight"] = _lin(g, embed_dim, embed_dim)
    sd[pre + "attn.out_proj.bias"] = _vec(g, embed_dim)
    _norm(g, sd, pre + "ln_q", embed_dim)
    _norm(g, sd, pre + "ln_kv", embed_dim)
    return sd


def vit_sd(cfg, seed=1234):
    g = _g(seed)
    W, od = cfg["width"], cfg["output_dim"]
    mlp = int(W * cfg["mlp_ratio"])
    sd = {}
    sd["positional_embedding"] = torch.randn(256, W, generator=g) * 0.3
    sd["proj"] = _lin(g, od, od)
    sd["conv1.weight"] = torch.randn(W, 3, cfg["patch_size"], cfg["patch_size"], generator=g) / math.sqrt(3 * cfg["patch_size"] ** 2)
    _norm(g, sd, "ln_pre", W)
    for i in range(cfg["layers"]):
        p = f"transformer.resblocks.{i}."
        _norm(g, sd, p + "ln_1", W)
        _norm(g, sd, p + "ln_2", W)
        sd[p + "attn.in_proj.weight"] = _lin(g, 3 * W, W, 1.5)
        sd[p + "attn.in_proj.bias"] = _vec(g, 3 * W)
        sd[p + "attn.out_proj.weight"] = _lin(g, W, W, 0.5)
        sd[p + "attn.out_proj.bias"] = _vec(g, W)
        sd[p + "mlp.c_fc.weight"] = _lin(g, mlp, W)
        sd[p + "mlp.c_fc.bias"] = _vec(g, mlp)
        sd[p + "mlp.c_proj.weight"] = _lin(g, W, mlp, 0.5)
        sd[p + "mlp.c_proj.bias"] = _vec(g, W)
    resampler_sd(g, "attn_pool.", int(math.sqrt(cfg["n_queries"])), od, W, sd)
    _norm(g, sd, "ln_post", od)
    return sd


def llama_sd(cfg, seed=1234):
    g = _g(seed + 1)
    H, I, V = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"]
    sd = {"model.embed_tokens.weight": torch.randn(V, H, generator=g) * 0.5}
    for i in range(cfg["num_hidden_layers"]):
        p = f"model.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj"):
            sd[p + f"self_attn.{n}.weight"] = _lin(g, H, H, 1.5)
        sd[p + "self_attn.o_proj.weight"] = _lin(g, H, H, 0.5)
        sd[p + "mlp.gate_proj.weight"] = _lin(g, I, H)
        sd[p + "mlp.up_proj.weight"] = _lin(g, I, H)
        sd[p + "mlp.down_proj.weight"] = _lin(g, H, I, 0.5)
        sd[p + "input_layernorm.weight"] = 1.0 + 0.1 * torch.randn(H, generator=g)
        sd[p + "post_attention_layernorm.weight"] = 1.0 + 0.1 * torch.randn(H, generator=g)
    sd["model.norm.weight"] = 1.0 + 0.1 * torch.randn(H, generator=g)
    sd["lm_head.weight"] = _lin(g, V, H, 2.0)
    return sd


def agent_sd(llm_cfg, vit_out_dim, in_grid=8, out_grid=8, seed=1234):
    """ContinuousLVLM extras: input/output resamplers + patch_pos_embed (seed_x.py:22-46, agent_seed_x_i.yaml)."""
    g = _g(seed + 2)
    H = llm_cfg["hidden_size"]
    sd = {}
    resampler_sd(g, "input_resampler.", in_grid, H, vit_out_dim, sd)
    resampler_sd(g, "output_resampler.", out_grid, vit_out_dim, H, sd)
    sd["patch_pos_embed"] = torch.randn(4, H, generator=g) * (H ** -0.5)
    return sd


def xlv2_sd(cfg, seed=1234, pre="resampler."):
    g = _g(seed + 3)
    dim, inner = cfg["dim"], cfg["dim_head"] * cfg["heads"]
    sd = {pre + "latents": torch.randn(1, cfg["num_queries"], dim, generator=g) * 0.5}
    sd[pre + "proj_in.weight"] = _lin(g, dim, cfg["embedding_dim"])
    sd[pre + "proj_in.bias"] = _vec(g, dim)
    _norm(g, sd, pre + "norm_out", dim)
    for i in range(cfg["depth"]):
        a, f = f"{pre}layers.{i}.0.", f"{pre}layers.{i}.1."
        _norm(g, sd, a + "norm1", dim)
        _norm(g, sd, a + "norm2", dim)
        sd[a + "to_q.weight"] = _lin(g, inner, dim, 1.5)
        sd[a + "to_kv.weight"] = _lin(g, 2 * inner, dim, 1.5)
        sd[a + "to_out.weight"] = _lin(g, dim, inner, 0.5)
        _norm(g, sd, f + "0", dim)
        sd[f + "1.weight"] = _lin(g, dim * cfg["ff_mult"], dim)
        sd[f + "3.weight"] = _lin(g, dim, dim * cfg["ff_mult"], 0.5)
    sd[pre + "unet_proj_1.weight"] = _lin(g, cfg["output1_dim"], dim)
    sd[pre + "unet_proj_1.bias"] = _vec(g, cfg["output1_dim"])
    sd[pre + "unet_proj_2.weight"] = _lin(g, cfg["output2_dim"], dim)
    sd[pre + "unet_proj_2.bias"] = _vec(g, cfg["output2_dim"])
    p = pre + "unet_attnpool."
    sd[p + "positional_embedding"] = torch.randn(cfg["num_queries"] + 1, dim, generator=g) / dim ** 0.5
    for n in ("k_proj", "q_proj", "v_proj"):
        sd[p + n + ".weight"] = _lin(g, dim, dim, 1.5)
        sd[p + n + ".bias"] = _vec(g, dim)
    sd[p + "c_proj.weight"] = _lin(g, cfg["output2_dim"], dim)
    sd[p + "c_proj.bias"] = _vec(g, cfg["output2_dim"])
    return sd
