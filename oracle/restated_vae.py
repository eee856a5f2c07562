"""CPU (fp32 torch) restatement of the SDXL VAE (decoder + encoder) — TEST INFRASTRUCTURE, **parity unpinned**.

``AutoencoderKL`` belongs to third-party **diffusers==0.25.0** (reference requirements.txt:4), which is neither under
/root/reference nor installed here. The decoder is restated from the published architecture with diffusers' state-dict key
names, anchored on the reference call site
(pipeline_stable_diffusion_xl_t2i_edit.py:965-977: ``vae.decode(latents / vae.config.scaling_factor)[0]`` after the fp32
upcast of ``upcast_vae`` :569-586) and checked by the exact parameter count of the SDXL VAE decoder config
(decoder 49 490 179 + post_quant_conv 20; encoder 34 163 592 + quant_conv 72; whole AutoencoderKL 83 653 863). The
encoder serves the edit pipeline's ``vae.encode(image).latent_dist.mode()`` (:505-527): mode() = the mean half of the
moments, no scaling_factor there.

SDXL vae/config.json: block_out_channels (128, 256, 512, 512), layers_per_block 2 (decoder: +1 resnet per up block),
latent_channels 4, norm_num_groups 32, act silu, one mid-block attention with a single 512-wide head (GroupNorm in front,
residual connection, eps 1e-6 everywhere), scaling_factor 0.13025, force_upcast true.
"""
import math

import torch
import torch.nn.functional as F

FULL_VAE = dict(latent_channels=4, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                norm_groups=32, scaling_factor=0.13025)
MINI_VAE = dict(latent_channels=4, out_channels=3, block_out_channels=(64, 128), layers_per_block=1, norm_groups=32,
                scaling_factor=0.13025)
EPS = 1e-6


def vae_decoder_param_shapes(cfg):
    """Ordered {name: shape} with diffusers 0.25.0 AutoencoderKL key names (decoder + post_quant_conv)."""
    boc = cfg["block_out_channels"]
    S = {}

    def conv(n, co, ci, k):
        S[n + ".weight"] = (co, ci, k, k)
        S[n + ".bias"] = (co,)

    def lin(n, o, i):
        S[n + ".weight"] = (o, i)
        S[n + ".bias"] = (o,)

    def norm(n, c):
        S[n + ".weight"] = (c,)
        S[n + ".bias"] = (c,)

    def resnet(n, ci, co):
        norm(n + ".norm1", ci)
        conv(n + ".conv1", co, ci, 3)
        norm(n + ".norm2", co)
        conv(n + ".conv2", co, co, 3)
        if ci != co:
            conv(n + ".conv_shortcut", co, ci, 1)

    lc = cfg["latent_channels"]
    conv("post_quant_conv", lc, lc, 1)
    top = boc[-1]
    conv("decoder.conv_in", top, lc, 3)
    resnet("decoder.mid_block.resnets.0", top, top)
    a = "decoder.mid_block.attentions.0"
    norm(a + ".group_norm", top)
    for nm in ("to_q", "to_k", "to_v"):
        lin(a + "." + nm, top, top)
    lin(a + ".to_out.0", top, top)
    resnet("decoder.mid_block.resnets.1", top, top)
    rev = list(reversed(boc))
    prev = rev[0]
    for i, co in enumerate(rev):
        for j in range(cfg["layers_per_block"] + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else co, co)
        prev = co
        if i != len(rev) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", co, co, 3)
    norm("decoder.conv_norm_out", boc[0])
    conv("decoder.conv_out", cfg["out_channels"], boc[0], 3)
    return S


def vae_decoder_param_count(cfg):
    return sum(math.prod(s) for s in vae_decoder_param_shapes(cfg).values())


def vae_sd(cfg, seed=1234, device="cpu", dtype=torch.float32):
    """Random-init state dict: conv/linear ~ N(0, 1/fan_in) (activations stay O(1) through the 30-odd layers), norm
    weights 1 + 0.1 N, biases 0.02 N."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    sd = {}
    for n, shp in vae_decoder_param_shapes(cfg).items():
        if n.endswith(".bias"):
            t = torch.randn(shp, generator=g) * 0.02
        elif len(shp) == 1:
            t = 1.0 + 0.1 * torch.randn(shp, generator=g)
        else:
            t = torch.randn(shp, generator=g) / math.sqrt(math.prod(shp[1:]))
        sd[n] = t.to(device=device, dtype=dtype)
    return sd


def _gn(sd, n, x, groups):
    return F.group_norm(x, groups, sd[n + ".weight"], sd[n + ".bias"], eps=EPS)


def _resnet(sd, n, x, groups):
    """ResnetBlock2D with temb=None, output_scale_factor 1 (diffusers resnet.py)."""
    h = F.conv2d(F.silu(_gn(sd, n + ".norm1", x, groups)), sd[n + ".conv1.weight"], sd[n + ".conv1.bias"], padding=1)
    h = F.conv2d(F.silu(_gn(sd, n + ".norm2", h, groups)), sd[n + ".conv2.weight"], sd[n + ".conv2.bias"], padding=1)
    if n + ".conv_shortcut.weight" in sd:
        x = F.conv2d(x, sd[n + ".conv_shortcut.weight"], sd[n + ".conv_shortcut.bias"])
    return x + h


def _mid_attention(sd, n, x, groups):
    """Attention(heads=1, dim_head=C, residual_connection=True, norm_num_groups=groups): tokens = pixels."""
    B, C, H, W = x.shape
    t = _gn(sd, n + ".group_norm", x, groups).view(B, C, H * W).transpose(1, 2)          # [B, HW, C]
    q = F.linear(t, sd[n + ".to_q.weight"], sd[n + ".to_q.bias"])
    k = F.linear(t, sd[n + ".to_k.weight"], sd[n + ".to_k.bias"])
    v = F.linear(t, sd[n + ".to_v.weight"], sd[n + ".to_v.bias"])
    p = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(C), dim=-1)
    o = F.linear(p @ v, sd[n + ".to_out.0.weight"], sd[n + ".to_out.0.bias"])
    return x + o.transpose(1, 2).reshape(B, C, H, W)


def vae_decode(sd, cfg, z):
    """AutoencoderKL.decode(z).sample: post_quant_conv → Decoder. z: [B, latent, h, w] (ALREADY divided by
    scaling_factor, as the reference call site does) → [B, 3, h·2^(nb-1), w·2^(nb-1)] fp32."""
    g = cfg["norm_groups"]
    x = F.conv2d(z.float(), sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    x = F.conv2d(x, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], padding=1)
    x = _resnet(sd, "decoder.mid_block.resnets.0", x, g)
    x = _mid_attention(sd, "decoder.mid_block.attentions.0", x, g)
    x = _resnet(sd, "decoder.mid_block.resnets.1", x, g)
    nb = len(cfg["block_out_channels"])
    for i in range(nb):
        for j in range(cfg["layers_per_block"] + 1):
            x = _resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}", x, g)
        if i != nb - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = F.conv2d(x, sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"],
                         sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"], padding=1)
    x = F.silu(_gn(sd, "decoder.conv_norm_out", x, g))
    return F.conv2d(x, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=1)


# ---------------------------------------------------------------------------------------------------------
# encoder (edit pipeline: image → latent_dist.mode())
# ---------------------------------------------------------------------------------------------------------
def vae_encoder_param_shapes(cfg, in_channels=3):
    boc = cfg["block_out_channels"]
    S = {}

    def conv(n, co, ci, k):
        S[n + ".weight"] = (co, ci, k, k)
        S[n + ".bias"] = (co,)

    def lin(n, o, i):
        S[n + ".weight"] = (o, i)
        S[n + ".bias"] = (o,)

    def norm(n, c):
        S[n + ".weight"] = (c,)
        S[n + ".bias"] = (c,)

    def resnet(n, ci, co):
        norm(n + ".norm1", ci)
        conv(n + ".conv1", co, ci, 3)
        norm(n + ".norm2", co)
        conv(n + ".conv2", co, co, 3)
        if ci != co:
            conv(n + ".conv_shortcut", co, ci, 1)

    lc = cfg["latent_channels"]
    conv("encoder.conv_in", boc[0], in_channels, 3)
    prev = boc[0]
    for i, co in enumerate(boc):
        for j in range(cfg["layers_per_block"]):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}", prev if j == 0 else co, co)
        prev = co
        if i != len(boc) - 1:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", co, co, 3)
    top = boc[-1]
    resnet("encoder.mid_block.resnets.0", top, top)
    a = "encoder.mid_block.attentions.0"
    norm(a + ".group_norm", top)
    for nm in ("to_q", "to_k", "to_v"):
        lin(a + "." + nm, top, top)
    lin(a + ".to_out.0", top, top)
    resnet("encoder.mid_block.resnets.1", top, top)
    norm("encoder.conv_norm_out", top)
    conv("encoder.conv_out", 2 * lc, top, 3)
    conv("quant_conv", 2 * lc, 2 * lc, 1)
    return S


def vae_encoder_param_count(cfg):
    return sum(math.prod(s) for s in vae_encoder_param_shapes(cfg).values())


def vae_encoder_sd(cfg, seed=1235, device="cpu", dtype=torch.float32):
    g = torch.Generator(device="cpu").manual_seed(seed)
    sd = {}
    for n, shp in vae_encoder_param_shapes(cfg).items():
        if n.endswith(".bias"):
            t = torch.randn(shp, generator=g) * 0.02
        elif len(shp) == 1:
            t = 1.0 + 0.1 * torch.randn(shp, generator=g)
        else:
            t = torch.randn(shp, generator=g) / math.sqrt(math.prod(shp[1:]))
        sd[n] = t.to(device=device, dtype=dtype)
    return sd


def vae_encode_mode(sd, cfg, image):
    """AutoencoderKL.encode(image).latent_dist.mode(): Encoder → quant_conv → mean half. image: [B, 3, H, W] in [-1, 1]
    → [B, latent, H/8, W/8]. Downsample2D of the encoder pads (0,1,0,1) and convolves with stride 2, padding 0."""
    g = cfg["norm_groups"]
    x = F.conv2d(image.float(), sd["encoder.conv_in.weight"], sd["encoder.conv_in.bias"], padding=1)
    nb = len(cfg["block_out_channels"])
    for i in range(nb):
        for j in range(cfg["layers_per_block"]):
            x = _resnet(sd, f"encoder.down_blocks.{i}.resnets.{j}", x, g)
        if i != nb - 1:
            x = F.pad(x, (0, 1, 0, 1))
            x = F.conv2d(x, sd[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"],
                         sd[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"], stride=2)
    x = _resnet(sd, "encoder.mid_block.resnets.0", x, g)
    x = _mid_attention(sd, "encoder.mid_block.attentions.0", x, g)
    x = _resnet(sd, "encoder.mid_block.resnets.1", x, g)
    x = F.silu(_gn(sd, "encoder.conv_norm_out", x, g))
    x = F.conv2d(x, sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"], padding=1)
    moments = F.conv2d(x, sd["quant_conv.weight"], sd["quant_conv.bias"])
    return moments[:, : cfg["latent_channels"]]
