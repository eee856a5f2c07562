"""CPU (fp32 torch) restatement of the de-tokenizer body — TEST INFRASTRUCTURE, **parity unpinned**.

The SDXL ``UNet2DConditionModel``, ``EulerDiscreteScheduler`` and the t2i ``StableDiffusionXLPipeline.__call__``
loop belong to third-party **diffusers==0.25.0** (reference requirements.txt:4, License_Seed-X.txt:80), which is
neither under /root/reference nor installed here. They are restated from the published architecture
 using diffusers' state-dict key names, anchored on the reference call sites
(adapter_modules.py:45,78-84,156-167; pipeline_stable_diffusion_xl_t2i_edit.py:915-922,953) and checked by the exact
parameter count of the SDXL-base config (2 567 463 684; 8-channel edit variant 2 567 475 204). The EDIT loop is
in-tree and followed line by line (pipeline_stable_diffusion_xl_t2i_edit.py:900-963).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

FULL_UNET = dict(in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280), layers_per_block=2,
                 down_attn=(False, True, True), up_attn=(True, True, False), transformer_layers=(1, 2, 10),
                 heads=(5, 10, 20), cross_attention_dim=2048, addition_time_embed_dim=256, pooled_dim=1280,
                 norm_groups=32)
MINI_UNET = dict(in_channels=4, out_channels=4, block_out_channels=(64, 128, 256), layers_per_block=2,
                 down_attn=(False, True, True), up_attn=(True, True, False), transformer_layers=(1, 1, 2),
                 heads=(1, 2, 4), cross_attention_dim=128, addition_time_embed_dim=32, pooled_dim=128,
                 norm_groups=32)


# ---------------------------------------------------------------------------------------------------------
# parameter inventory (shared by the weight generator and the HIP loader tests)
# ---------------------------------------------------------------------------------------------------------
def unet_param_shapes(cfg):
    """Ordered {name: shape} with diffusers 0.25.0 UNet2DConditionModel key names for an SDXL-style config."""
    boc = cfg["block_out_channels"]
    ted = boc[0] * 4
    ca = cfg["cross_attention_dim"]
    S = {}

    def conv(n, co, ci, k):
        S[n + ".weight"] = (co, ci, k, k)
        S[n + ".bias"] = (co,)

    def lin(n, o, i, bias=True):
        S[n + ".weight"] = (o, i)
        if bias:
            S[n + ".bias"] = (o,)

    def norm(n, c):
        S[n + ".weight"] = (c,)
        S[n + ".bias"] = (c,)

    def resnet(n, ci, co):
        norm(n + ".norm1", ci)
        conv(n + ".conv1", co, ci, 3)
        lin(n + ".time_emb_proj", co, ted)
        norm(n + ".norm2", co)
        conv(n + ".conv2", co, co, 3)
        if ci != co:
            conv(n + ".conv_shortcut", co, ci, 1)

    def transformer(n, c, layers):
        norm(n + ".norm", c)
        lin(n + ".proj_in", c, c)
        for k in range(layers):
            b = f"{n}.transformer_blocks.{k}"
            norm(b + ".norm1", c)
            for q in ("to_q", "to_k", "to_v"):
                lin(f"{b}.attn1.{q}", c, c, bias=False)
            lin(b + ".attn1.to_out.0", c, c)
            norm(b + ".norm2", c)
            lin(b + ".attn2.to_q", c, c, bias=False)
            lin(b + ".attn2.to_k", c, ca, bias=False)
            lin(b + ".attn2.to_v", c, ca, bias=False)
            lin(b + ".attn2.to_out.0", c, c)
            norm(b + ".norm3", c)
            lin(b + ".ff.net.0.proj", 8 * c, c)
            lin(b + ".ff.net.2", c, 4 * c)
        lin(n + ".proj_out", c, c)

    conv("conv_in", boc[0], cfg["in_channels"], 3)
    lin("time_embedding.linear_1", ted, boc[0])
    lin("time_embedding.linear_2", ted, ted)
    lin("add_embedding.linear_1", ted, 6 * cfg["addition_time_embed_dim"] + cfg["pooled_dim"])
    lin("add_embedding.linear_2", ted, ted)
    out_c = boc[0]
    for i, co in enumerate(boc):
        ci, out_c = out_c, co
        for j in range(cfg["layers_per_block"]):
            resnet(f"down_blocks.{i}.resnets.{j}", ci if j == 0 else co, co)
            if cfg["down_attn"][i]:
                transformer(f"down_blocks.{i}.attentions.{j}", co, cfg["transformer_layers"][i])
        if i != len(boc) - 1:
            conv(f"down_blocks.{i}.downsamplers.0.conv", co, co, 3)
    cm = boc[-1]
    resnet("mid_block.resnets.0", cm, cm)
    transformer("mid_block.attentions.0", cm, cfg["transformer_layers"][-1])
    resnet("mid_block.resnets.1", cm, cm)
    rev = list(reversed(boc))
    rev_layers = list(reversed(cfg["transformer_layers"]))
    out_c = rev[0]
    for i, co in enumerate(rev):
        prev, out_c = out_c, co
        in_c = rev[min(i + 1, len(boc) - 1)]
        n_res = cfg["layers_per_block"] + 1
        for j in range(n_res):
            skip = in_c if j == n_res - 1 else co
            rin = prev if j == 0 else co
            resnet(f"up_blocks.{i}.resnets.{j}", rin + skip, co)
            if cfg["up_attn"][i]:
                transformer(f"up_blocks.{i}.attentions.{j}", co, rev_layers[i])
        if i != len(boc) - 1:
            conv(f"up_blocks.{i}.upsamplers.0.conv", co, co, 3)
    norm("conv_norm_out", boc[0])
    conv("conv_out", cfg["out_channels"], boc[0], 3)
    return S


def unet_param_count(cfg):
    return sum(int(np.prod(s)) for s in unet_param_shapes(cfg).values())


def unet_sd(cfg, seed=1234, device="cpu", dtype=torch.float32):
    """Seeded random UNet weights. Scales keep activations O(1) through ~70 residual blocks: fan-in normalised
    projections with a small gain on every residual-branch output."""
    g = torch.Generator(device=device).manual_seed(seed + 4)
    sd = {}
    for name, shape in unet_param_shapes(cfg).items():
        if name.endswith(".bias"):
            is_norm = any(t in name for t in (".norm", "conv_norm_out"))
            w = torch.randn(shape, generator=g, device=device) * (0.05 if is_norm else 0.02)
        elif len(shape) == 1:
            w = 1.0 + 0.1 * torch.randn(shape, generator=g, device=device)
        else:
            fan_in = int(np.prod(shape[1:]))
            gain = 1.0
            if any(t in name for t in ("conv2.", "to_out.0", "ff.net.2", "proj_out")):
                gain = 0.4                          # residual-branch outputs
            if any(t in name for t in ("to_q", "to_k")):
                gain = 1.3
            w = torch.randn(shape, generator=g, device=device) * (gain / math.sqrt(fan_in))
        sd[name] = w.to(dtype)
    return sd


# ---------------------------------------------------------------------------------------------------------
# UNet forward
# ---------------------------------------------------------------------------------------------------------
def timestep_embedding(t, dim):
    """diffusers get_timestep_embedding with flip_sin_to_cos=True, downscale_freq_shift=0 [ext]."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    arg = t.float()[:, None] * freqs[None]
    return torch.cat([arg.cos(), arg.sin()], dim=-1)


def _resnet(sd, n, x, emb, groups):
    h = F.silu(F.group_norm(x, groups, sd[n + ".norm1.weight"], sd[n + ".norm1.bias"], 1e-5))
    h = F.conv2d(h, sd[n + ".conv1.weight"], sd[n + ".conv1.bias"], padding=1)
    t = F.linear(F.silu(emb), sd[n + ".time_emb_proj.weight"], sd[n + ".time_emb_proj.bias"])
    h = h + t[:, :, None, None]
    h = F.silu(F.group_norm(h, groups, sd[n + ".norm2.weight"], sd[n + ".norm2.bias"], 1e-5))
    h = F.conv2d(h, sd[n + ".conv2.weight"], sd[n + ".conv2.bias"], padding=1)
    if n + ".conv_shortcut.weight" in sd:
        x = F.conv2d(x, sd[n + ".conv_shortcut.weight"], sd[n + ".conv_shortcut.bias"])
    return x + h


def _attn(sd, n, x, ctx, heads):
    q = F.linear(x, sd[n + ".to_q.weight"])
    k = F.linear(ctx, sd[n + ".to_k.weight"])
    v = F.linear(ctx, sd[n + ".to_v.weight"])
    B, L, C = q.shape
    hd = C // heads

    def sp(t):
        return t.view(B, -1, heads, hd).transpose(1, 2)
    a = torch.softmax(sp(q) @ sp(k).transpose(-1, -2) * hd ** -0.5, dim=-1) @ sp(v)
    a = a.transpose(1, 2).reshape(B, L, C)
    return F.linear(a, sd[n + ".to_out.0.weight"], sd[n + ".to_out.0.bias"])


def _transformer(sd, n, x, ehs, heads, layers, groups):
    B, C, H, W = x.shape
    res = x
    h = F.group_norm(x, groups, sd[n + ".norm.weight"], sd[n + ".norm.bias"], 1e-6)
    h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    h = F.linear(h, sd[n + ".proj_in.weight"], sd[n + ".proj_in.bias"])
    for k in range(layers):
        b = f"{n}.transformer_blocks.{k}"
        y = F.layer_norm(h, (C,), sd[b + ".norm1.weight"], sd[b + ".norm1.bias"], 1e-5)
        h = _attn(sd, b + ".attn1", y, y, heads) + h
        y = F.layer_norm(h, (C,), sd[b + ".norm2.weight"], sd[b + ".norm2.bias"], 1e-5)
        h = _attn(sd, b + ".attn2", y, ehs, heads) + h
        y = F.layer_norm(h, (C,), sd[b + ".norm3.weight"], sd[b + ".norm3.bias"], 1e-5)
        p = F.linear(y, sd[b + ".ff.net.0.proj.weight"], sd[b + ".ff.net.0.proj.bias"])
        hid, gate = p.chunk(2, dim=-1)
        h = F.linear(hid * F.gelu(gate), sd[b + ".ff.net.2.weight"], sd[b + ".ff.net.2.bias"]) + h
    h = F.linear(h, sd[n + ".proj_out.weight"], sd[n + ".proj_out.bias"])
    return h.reshape(B, H, W, C).permute(0, 3, 1, 2) + res


def unet_forward(sd, cfg, sample, timestep, encoder_hidden_states, text_embeds, time_ids):
    """UNet2DConditionModel.forward [ext]: sample [B,Cin,H,W], timestep scalar/[B], ehs [B,L,ca], text_embeds
    [B,pooled], time_ids [B,6] → [B,Cout,H,W]."""
    boc, G = cfg["block_out_channels"], cfg["norm_groups"]
    B = sample.shape[0]
    t = torch.as_tensor(timestep, dtype=torch.float32).reshape(-1).expand(B).to(sample.device)
    emb = timestep_embedding(t, boc[0])
    emb = F.linear(F.silu(F.linear(emb, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])),
                   sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])
    te = timestep_embedding(time_ids.flatten(), cfg["addition_time_embed_dim"]).reshape(B, -1)
    add = torch.cat([text_embeds.float(), te], dim=-1)
    aug = F.linear(F.silu(F.linear(add, sd["add_embedding.linear_1.weight"], sd["add_embedding.linear_1.bias"])),
                   sd["add_embedding.linear_2.weight"], sd["add_embedding.linear_2.bias"])
    emb = emb + aug
    ehs = encoder_hidden_states.float()
    x = F.conv2d(sample.float(), sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
    skips = [x]
    for i in range(len(boc)):
        for j in range(cfg["layers_per_block"]):
            x = _resnet(sd, f"down_blocks.{i}.resnets.{j}", x, emb, G)
            if cfg["down_attn"][i]:
                x = _transformer(sd, f"down_blocks.{i}.attentions.{j}", x, ehs, cfg["heads"][i],
                                 cfg["transformer_layers"][i], G)
            skips.append(x)
        if i != len(boc) - 1:
            n = f"down_blocks.{i}.downsamplers.0.conv"
            x = F.conv2d(x, sd[n + ".weight"], sd[n + ".bias"], stride=2, padding=1)
            skips.append(x)
    x = _resnet(sd, "mid_block.resnets.0", x, emb, G)
    x = _transformer(sd, "mid_block.attentions.0", x, ehs, cfg["heads"][-1], cfg["transformer_layers"][-1], G)
    x = _resnet(sd, "mid_block.resnets.1", x, emb, G)
    rev_heads = list(reversed(cfg["heads"]))
    rev_layers = list(reversed(cfg["transformer_layers"]))
    for i in range(len(boc)):
        for j in range(cfg["layers_per_block"] + 1):
            x = torch.cat([x, skips.pop()], dim=1)
            x = _resnet(sd, f"up_blocks.{i}.resnets.{j}", x, emb, G)
            if cfg["up_attn"][i]:
                x = _transformer(sd, f"up_blocks.{i}.attentions.{j}", x, ehs, rev_heads[i], rev_layers[i], G)
        if i != len(boc) - 1:
            n = f"up_blocks.{i}.upsamplers.0.conv"
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = F.conv2d(x, sd[n + ".weight"], sd[n + ".bias"], padding=1)
    x = F.silu(F.group_norm(x, G, sd["conv_norm_out.weight"], sd["conv_norm_out.bias"], 1e-5))
    return F.conv2d(x, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)


# ---------------------------------------------------------------------------------------------------------
# EulerDiscreteScheduler (SDXL scheduler config) + CFG loops
# ---------------------------------------------------------------------------------------------------------
def euler_tables(num_inference_steps, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1):
    """scaled_linear betas, `leading` spacing, steps_offset 1, linear sigma interpolation, final sigma 0 [ext].
    Returns (timesteps float32 [N], sigmas float32 [N+1], init_noise_sigma)."""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32, device="cpu") ** 2
    ac = torch.cumprod(1.0 - betas, dim=0).numpy()
    step_ratio = num_train_timesteps // num_inference_steps
    ts = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.float32) + steps_offset
    sig = np.array(((1 - ac) / ac) ** 0.5)
    sig = np.interp(ts, np.arange(0, len(sig)), sig)
    sig = np.concatenate([sig, [0.0]]).astype(np.float32)
    init = float((sig.max() ** 2 + 1) ** 0.5)
    return torch.from_numpy(ts), torch.from_numpy(sig), init


def t2i_loop(unet_fn, latents, prompt_embeds, neg_prompt_embeds, pooled, neg_pooled, time_ids, num_steps,
             guidance_scale=7.5):
    """StableDiffusionXLPipeline.__call__ denoise loop [ext] as driven by adapter_modules.py:156-167: order
    [uncond, text]; latents must already be scaled by init_noise_sigma. unet_fn(sample, t, ehs, text_embeds, time_ids)."""
    ts, sig, _ = euler_tables(num_steps)
    ehs = torch.cat([neg_prompt_embeds, prompt_embeds], dim=0)
    te = torch.cat([neg_pooled, pooled], dim=0)
    tid = torch.cat([time_ids, time_ids], dim=0)
    lat = latents.float().clone()
    for i in range(num_steps):
        s = float(sig[i])
        inp = torch.cat([lat] * 2) / ((s ** 2 + 1) ** 0.5)
        eps = unet_fn(inp, ts[i], ehs, te, tid)
        eu, et = eps.chunk(2)
        e = eu + guidance_scale * (et - eu)
        lat = lat + e * (float(sig[i + 1]) - s)
    return lat


def edit_loop(unet_fn, latents, image_latents, prompt_embeds, neg_prompt_embeds, pooled, neg_pooled, time_ids,
              num_steps, guidance_scale=7.5, image_guidance_scale=1.5):
    """pipeline_stable_diffusion_xl_t2i_edit.py:884-886,900-963: order [text, image, uncond]; image_latents
    [3,4,H,W] = [enc, enc, 0] (:544-546) concatenated on channels (:911); sigma-space guidance hack (:928-950)."""
    ts, sig, _ = euler_tables(num_steps)
    ehs = torch.cat([prompt_embeds, neg_prompt_embeds, neg_prompt_embeds], dim=0)
    te = torch.cat([pooled, neg_pooled, neg_pooled], dim=0)
    tid = torch.cat([time_ids] * 3, dim=0)
    lat = latents.float().clone()
    for i in range(num_steps):
        s = float(sig[i])
        lmi = torch.cat([lat] * 3)
        inp = torch.cat([lmi / ((s ** 2 + 1) ** 0.5), image_latents], dim=1)
        npred = unet_fn(inp, ts[i], ehs, te, tid)
        npred = lmi - s * npred                                              # :931
        nt, ni, nu = npred.chunk(3)
        npred = nu + guidance_scale * (nt - ni) + image_guidance_scale * (ni - nu)   # :935-937
        npred = (npred - lat) / (-s)                                          # :950
        lat = lat + npred * (float(sig[i + 1]) - s)                           # :953 (Euler step)
    return lat
