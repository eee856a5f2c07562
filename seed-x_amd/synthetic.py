"""Synthetic (seeded random) weights of the SEED-X architecture, generated directly in HBM.

There are no checkpoints and no network in the build/bench environment, so ``bench.py`` and the
full-dimension smoke tests run the real architecture with random-init weights. Tensors use the reference's
state-dict key names, so they go through exactly the same ``load_state_dict`` → pack path as a real checkpoint.
Scales keep activations O(1) through the depth of each network.
"""
import math

import torch

FULL_VIT = dict(image_size=448, patch_size=14, width=1664, layers=48, heads=16, mlp_ratio=4.9231, n_queries=256,
                output_dim=4096)                                   # configs/visual_encoder/qwen_vitg_448.yaml
FULL_LLM = dict(hidden_size=5120, intermediate_size=13824, num_hidden_layers=40, num_attention_heads=40,
                vocab_size=32330, rms_norm_eps=1e-5, max_position_embeddings=4096)   # Llama-2-13B dims
FULL_XLV2 = dict(dim=1024, depth=4, dim_head=64, heads=16, num_queries=64, embedding_dim=4096, output1_dim=768,
                 output2_dim=1280, ff_mult=4)                       # configs/sdxl_adapter/*.yaml


def unet_param_shapes(cfg):
    """{name: shape} of a diffusers-0.25.0 SDXL-style UNet2DConditionModel (key layout in SURVEY.md §8c)."""
    boc = cfg["block_out_channels"]
    ted, ca = boc[0] * 4, cfg["cross_attention_dim"]
    S = {}

    def wb(n, *shape, bias=True):
        S[n + ".weight"] = tuple(shape)
        if bias:
            S[n + ".bias"] = (shape[0],)

    def resnet(n, ci, co):
        wb(n + ".norm1", ci)
        wb(n + ".conv1", co, ci, 3, 3)
        wb(n + ".time_emb_proj", co, ted)
        wb(n + ".norm2", co)
        wb(n + ".conv2", co, co, 3, 3)
        if ci != co:
            wb(n + ".conv_shortcut", co, ci, 1, 1)

    def transformer(n, c, layers):
        wb(n + ".norm", c)
        wb(n + ".proj_in", c, c)
        for k in range(layers):
            b = f"{n}.transformer_blocks.{k}"
            wb(b + ".norm1", c)
            for q in ("to_q", "to_k", "to_v"):
                wb(f"{b}.attn1.{q}", c, c, bias=False)
            wb(b + ".attn1.to_out.0", c, c)
            wb(b + ".norm2", c)
            wb(b + ".attn2.to_q", c, c, bias=False)
            wb(b + ".attn2.to_k", c, ca, bias=False)
            wb(b + ".attn2.to_v", c, ca, bias=False)
            wb(b + ".attn2.to_out.0", c, c)
            wb(b + ".norm3", c)
            wb(b + ".ff.net.0.proj", 8 * c, c)
            wb(b + ".ff.net.2", c, 4 * c)
        wb(n + ".proj_out", c, c)

    wb("conv_in", boc[0], cfg["in_channels"], 3, 3)
    wb("time_embedding.linear_1", ted, boc[0])
    wb("time_embedding.linear_2", ted, ted)
    wb("add_embedding.linear_1", ted, 6 * cfg["addition_time_embed_dim"] + cfg["pooled_dim"])
    wb("add_embedding.linear_2", ted, ted)
    prev = boc[0]
    for i, co in enumerate(boc):
        for j in range(cfg["layers_per_block"]):
            resnet(f"down_blocks.{i}.resnets.{j}", prev if j == 0 else co, co)
            if cfg["down_attn"][i]:
                transformer(f"down_blocks.{i}.attentions.{j}", co, cfg["transformer_layers"][i])
        if i != len(boc) - 1:
            wb(f"down_blocks.{i}.downsamplers.0.conv", co, co, 3, 3)
        prev = co
    cm = boc[-1]
    resnet("mid_block.resnets.0", cm, cm)
    transformer("mid_block.attentions.0", cm, cfg["transformer_layers"][-1])
    resnet("mid_block.resnets.1", cm, cm)
    rev, rl = list(reversed(boc)), list(reversed(cfg["transformer_layers"]))
    prev = rev[0]
    for i, co in enumerate(rev):
        skip_last = rev[min(i + 1, len(boc) - 1)]
        nres = cfg["layers_per_block"] + 1
        for j in range(nres):
            skip = skip_last if j == nres - 1 else co
            resnet(f"up_blocks.{i}.resnets.{j}", (prev if j == 0 else co) + skip, co)
            if cfg["up_attn"][i]:
                transformer(f"up_blocks.{i}.attentions.{j}", co, rl[i])
        if i != len(boc) - 1:
            wb(f"up_blocks.{i}.upsamplers.0.conv", co, co, 3, 3)
        prev = co
    wb("conv_norm_out", boc[0])
    wb("conv_out", cfg["out_channels"], boc[0], 3, 3)
    return S


class _Gen:
    def __init__(self, device, dtype, seed):
        self.device, self.dtype = device, dtype
        self.g = torch.Generator(device=device).manual_seed(seed)

    def normal(self, shape, std):
        return (torch.randn(shape, generator=self.g, device=self.device, dtype=torch.float32) * std).to(self.dtype)

    def lin(self, o, i, gain=1.0):
        return self.normal((o, i), gain / math.sqrt(i))

    def gamma(self, n):
        return (1.0 + 0.1 * torch.randn(n, generator=self.g, device=self.device)).to(self.dtype)

    def vec(self, n, std=0.02):
        return self.normal((n,), std)


def _resampler(G, sd, pre, grid, E, kv):
    sd[pre + "query"] = G.normal((grid * grid, E), 0.5)
    if kv != E:
        sd[pre + "kv_proj.weight"] = G.lin(E, kv)
    sd[pre + "attn.in_proj_weight"] = G.lin(3 * E, E)
    sd[pre + "attn.in_proj_bias"] = G.vec(3 * E)
    sd[pre + "attn.out_proj.weight"] = G.lin(E, E)
    sd[pre + "attn.out_proj.bias"] = G.vec(E)
    for n in ("ln_q", "ln_kv"):
        sd[pre + n + ".weight"] = G.gamma(E)
        sd[pre + n + ".bias"] = G.vec(E, 0.05)


def vit_state_dict(cfg, device, dtype=torch.float16, seed=1234):
    G = _Gen(device, dtype, seed)
    W, od, ps = cfg["width"], cfg["output_dim"], cfg["patch_size"]
    mlp = int(W * cfg["mlp_ratio"])
    sd = {"positional_embedding": G.normal((256, W), 0.3), "proj": G.lin(od, od),
          "conv1.weight": G.normal((W, 3, ps, ps), 1.0 / math.sqrt(3 * ps * ps)),
          "ln_pre.weight": G.gamma(W), "ln_pre.bias": G.vec(W, 0.05),
          "ln_post.weight": G.gamma(od), "ln_post.bias": G.vec(od, 0.05)}
    for i in range(cfg["layers"]):
        p = f"transformer.resblocks.{i}."
        for n in ("ln_1", "ln_2"):
            sd[p + n + ".weight"], sd[p + n + ".bias"] = G.gamma(W), G.vec(W, 0.05)
        sd[p + "attn.in_proj.weight"], sd[p + "attn.in_proj.bias"] = G.lin(3 * W, W, 1.5), G.vec(3 * W)
        sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"] = G.lin(W, W, 0.5), G.vec(W)
        sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"] = G.lin(mlp, W), G.vec(mlp)
        sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"] = G.lin(W, mlp, 0.5), G.vec(W)
    _resampler(G, sd, "attn_pool.", int(math.sqrt(cfg["n_queries"])), od, W)
    return sd


def llama_state_dict(cfg, device, dtype=torch.float16, seed=1235):
    G = _Gen(device, dtype, seed)
    H, I, V = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"]
    sd = {"model.embed_tokens.weight": G.normal((V, H), 0.5), "model.norm.weight": G.gamma(H),
          "lm_head.weight": G.lin(V, H, 2.0)}
    for i in range(cfg["num_hidden_layers"]):
        p = f"model.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj"):
            sd[p + f"self_attn.{n}.weight"] = G.lin(H, H, 1.5)
        sd[p + "self_attn.o_proj.weight"] = G.lin(H, H, 0.5)
        sd[p + "mlp.gate_proj.weight"] = G.lin(I, H)
        sd[p + "mlp.up_proj.weight"] = G.lin(I, H)
        sd[p + "mlp.down_proj.weight"] = G.lin(H, I, 0.5)
        sd[p + "input_layernorm.weight"] = G.gamma(H)
        sd[p + "post_attention_layernorm.weight"] = G.gamma(H)
    return sd


def agent_state_dict(llm_hidden, vit_dim, device, dtype=torch.float16, seed=1236, grid=8):
    G = _Gen(device, dtype, seed)
    sd = {}
    _resampler(G, sd, "input_resampler.", grid, llm_hidden, vit_dim)
    _resampler(G, sd, "output_resampler.", grid, vit_dim, llm_hidden)
    sd["patch_pos_embed"] = G.normal((4, llm_hidden), llm_hidden ** -0.5)
    return sd


def xlv2_state_dict(cfg, device, dtype=torch.float16, seed=1237, pre="resampler."):
    G = _Gen(device, dtype, seed)
    dim, inner = cfg["dim"], cfg["dim_head"] * cfg["heads"]
    sd = {pre + "latents": G.normal((1, cfg["num_queries"], dim), 0.5),
          pre + "proj_in.weight": G.lin(dim, cfg["embedding_dim"]), pre + "proj_in.bias": G.vec(dim),
          pre + "norm_out.weight": G.gamma(dim), pre + "norm_out.bias": G.vec(dim, 0.05)}
    for i in range(cfg["depth"]):
        a, f = f"{pre}layers.{i}.0.", f"{pre}layers.{i}.1."
        for n in ("norm1", "norm2"):
            sd[a + n + ".weight"], sd[a + n + ".bias"] = G.gamma(dim), G.vec(dim, 0.05)
        sd[a + "to_q.weight"], sd[a + "to_kv.weight"] = G.lin(inner, dim, 1.5), G.lin(2 * inner, dim, 1.5)
        sd[a + "to_out.weight"] = G.lin(dim, inner, 0.5)
        sd[f + "0.weight"], sd[f + "0.bias"] = G.gamma(dim), G.vec(dim, 0.05)
        sd[f + "1.weight"], sd[f + "3.weight"] = G.lin(dim * cfg["ff_mult"], dim), G.lin(dim, dim * cfg["ff_mult"], 0.5)
    for n, o in (("unet_proj_1", cfg["output1_dim"]), ("unet_proj_2", cfg["output2_dim"])):
        sd[pre + n + ".weight"], sd[pre + n + ".bias"] = G.lin(o, dim), G.vec(o)
    p = pre + "unet_attnpool."
    sd[p + "positional_embedding"] = G.normal((cfg["num_queries"] + 1, dim), dim ** -0.5)
    for n in ("k_proj", "q_proj", "v_proj"):
        sd[p + n + ".weight"], sd[p + n + ".bias"] = G.lin(dim, dim, 1.5), G.vec(dim)
    sd[p + "c_proj.weight"], sd[p + "c_proj.bias"] = G.lin(cfg["output2_dim"], dim), G.vec(cfg["output2_dim"])
    return sd


def unet_state_dict(cfg, device, dtype=torch.float16, seed=1238):
    G = _Gen(device, dtype, seed)
    sd = {}
    for name, shape in unet_param_shapes(cfg).items():
        if name.endswith(".bias"):
            sd[name] = G.vec(shape[0], 0.05 if ".norm" in name or "conv_norm_out" in name else 0.02)
        elif len(shape) == 1:
            sd[name] = G.gamma(shape[0])
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            gain = 0.4 if any(t in name for t in ("conv2.", "to_out.0", "ff.net.2", "proj_out")) else \
                (1.3 if ("to_q" in name or "to_k" in name) else 1.0)
            sd[name] = G.normal(shape, gain / math.sqrt(fan_in))
    return sd


def vae_state_dict(vae, device, dtype=torch.float16, seed=1239):
    """Random-init decoder weights for a seedx_amd.vae.AutoencoderKL instance (its own key/shape inventory)."""
    G = _Gen(device, dtype, seed)
    sd = {}
    for name, shape in vae.param_shapes().items():
        if name.endswith(".bias"):
            sd[name] = G.vec(shape[0], 0.02)
        elif len(shape) == 1:
            sd[name] = G.gamma(shape[0])
        else:
            fan_in = 1
            for s_ in shape[1:]:
                fan_in *= s_
            sd[name] = G.normal(shape, (0.4 if "conv2." in name or "to_out.0" in name else 1.0) / math.sqrt(fan_in))
    return sd
