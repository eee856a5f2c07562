"""SDXL VAE decoder on the HIP kernels (the last stage of "image out").

Stand-in for ``diffusers.AutoencoderKL`` [ext, diffusers 0.25.0] as the reference uses it:
``image = vae.decode(latents / vae.config.scaling_factor, return_dict=False)[0]``
(pipeline_stable_diffusion_xl_t2i_edit.py:965-977, after ``upcast_vae`` :569-586 because the fp16 VAE overflows). Same
constructor config names, ``from_pretrained`` directory layout (``config.json`` + ``diffusion_pytorch_model.*``), state-dict
keys, ``.config.scaling_factor`` / ``.config.force_upcast`` / ``.dtype`` and ``decode()`` signature.

Design: NHWC end to end, fp32 residual stream, fp32 accumulation, MFMA operands in one of two precisions:

* ``fp32`` — the reference's pipeline upcasts an fp16 VAE with ``force_upcast`` to fp32 around every encode / decode
  (:509-511, :967-970). There is no fp32 MFMA worth using, so fp32-grade products are built from bf16 ones: every operand
  is carried as two bf16 planes x = hi + lo (sx_split_bf16) and A·W = Ah·Wh + Ah·Wl + Al·Wh comes out of ONE launch of the
  same GEMM / implicit-conv kernel over a tripled K: activation rows are laid out [hi | hi | lo] per pixel, weight rows
  [hi | lo | hi] per tap, all three products accumulate in the fp32 MFMA accumulators. 16 mantissa bits per operand, fp32's
  exponent range (the fp16 overflow that forces the reference into fp32 cannot occur). Selected exactly when the reference would upcast
  (dtype fp16 and ``config.force_upcast``), or by ``.to(dtype=torch.float32)``, or ``precision="fp32"``.
  Round 6: an fp16 VAE's parameters ARE fp16 values (``.to(dtype=torch.float16)`` rounded them; ``upcast_vae`` only widens them, and
  ``_pack`` does the same rounding), so a weight is ONE exact fp16 plane: the 3x3 convs behind a GroupNorm (+ SiLU: bounded outputs)
  — resnet conv1 / conv2, conv_out, 73 % of the decoder's conv time — take their activations as two fp16 planes [hi | lo]
  (GroupNorm output mode SX_F16X2) against weight rows duplicated per tap: A·W = Ah·W + Al·W, TWO products instead of three and 22
  activation mantissa bits instead of 16. The up-sampler convs read the un-normalised stream (the values that overflow fp16 in this
  VAE), the 1x1 shortcuts the raw block input, attention multiplies two activations: they keep the three bf16 planes.
* ``fast`` — single 16-bit operands of ``dtype`` (what a bf16 VAE is in the reference; ``precision="fast"`` forces it for
  fp16 too, ≈1.3e-3 rel-L2 from the fp32 result at 1024 px with fp16 operands). 3×3 convolutions = the implicit-GEMM kernel (nearest-2× upsample fused into the conv's gather), GroupNorm+SiLU
kernels feed it 16-bit operands. The mid block's single 512-wide attention head over (H/8)·(W/8) pixels does not fit the
flash kernel (head_dim ≤ 128): scores go through the GEMM in blocks of 2048 query rows ([2048, HW] fp32 = 128 MB at
1024 px instead of the 1-GiB full matrix), a row-softmax kernel emits 16-bit probabilities, V^T comes straight out of a GEMM with swapped operands (W_v · X^T), and V's bias is added after
P·V (softmax rows sum to 1). Images are decoded one at a time: at 1024 px one image already gives every launch ≥ 4096
tiles, and it keeps the operand descriptors under 2 GiB.

``encode(image).latent_dist.mode()`` (edit pipeline, :505-527) runs the encoder the same way (its stride-2 Downsample2D
convs pad only bottom/right: sx_gemm pad_mode 1); ``sample()`` of the posterior is not offered (the reference only calls
``mode()``). Encoder weights are optional in the state dict: without them ``encode`` raises.
"""
import json
import math
import os
from types import SimpleNamespace

import torch

from . import ops

EPS = 1e-6


def pack_planes(w32, taps=1, role="w"):
    """fp32 [N, taps*C] → bf16 [N, taps*3C]: per tap the planes of w = hi + lo (hi = bf16(w), lo = bf16(w − hi)) laid out
    [hi | lo | hi] (role "w": rows of a GEMM's W operand) or [hi | hi | lo] (role "a": a weight used as the A operand).
    Host-side twin of sx_split_bf16 for weights (pack time only); activations are split on the GPU."""
    w3 = w32.reshape(w32.shape[0], taps, -1)
    hi = w3.to(torch.bfloat16)
    lo = (w3 - hi.float()).to(torch.bfloat16)
    planes = [hi, lo, hi] if role == "w" else [hi, hi, lo]
    return torch.cat(planes, dim=2).reshape(w32.shape[0], -1).contiguous()


class DecoderOutput:
    def __init__(self, sample):
        self.sample = sample


class AutoencoderKL:
    def __init__(self, in_channels=3, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                 latent_channels=4, norm_num_groups=32, scaling_factor=0.13025, force_upcast=True, **_):
        self.config = SimpleNamespace(in_channels=in_channels, out_channels=out_channels,
                                      block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                                      latent_channels=latent_channels, norm_num_groups=norm_num_groups,
                                      scaling_factor=scaling_factor, force_upcast=force_upcast)
        self.device, self.dtype = None, torch.bfloat16
        self.precision = "auto"              # "auto": what the reference does for this dtype / force_upcast; "fast"; "fp32"
        self._sd, self._P = None, None
        self.has_encoder = False

    @property
    def split(self):
        """True when operands are carried as two bf16 planes (the fp32-grade mode)."""
        if self.precision == "auto":
            return self.dtype == torch.float32 or (self.dtype == torch.float16 and bool(self.config.force_upcast))
        return self.precision == "fp32"

    @property
    def operand_dtype(self):
        return torch.bfloat16 if (self.split or self.dtype == torch.float32) else self.dtype

    # ---- reference-compatible plumbing ---------------------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder=None, torch_dtype=None, **kw):
        d = os.path.join(pretrained_model_name_or_path, subfolder) if subfolder else pretrained_model_name_or_path
        m = cls(**json.load(open(os.path.join(d, "config.json"))))
        st = os.path.join(d, "diffusion_pytorch_model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(d, "diffusion_pytorch_model.bin"), map_location="cpu")
        m.load_state_dict(sd)
        if torch_dtype is not None:
            m.dtype = torch_dtype
        return m

    def param_shapes(self):
        """Ordered {state-dict key: shape} of what decode() needs (diffusers 0.25.0 names)."""
        c = self.config
        boc, top, lc = c.block_out_channels, c.block_out_channels[-1], c.latent_channels
        S = {}

        def conv(n, co, ci, k):
            S[n + ".weight"], S[n + ".bias"] = (co, ci, k, k), (co,)

        def vec(n, ch):
            S[n + ".weight"], S[n + ".bias"] = (ch,), (ch,)

        def resnet(n, ci, co):
            vec(n + ".norm1", ci)
            conv(n + ".conv1", co, ci, 3)
            vec(n + ".norm2", co)
            conv(n + ".conv2", co, co, 3)
            if ci != co:
                conv(n + ".conv_shortcut", co, ci, 1)
        conv("post_quant_conv", lc, lc, 1)
        conv("decoder.conv_in", top, lc, 3)
        resnet("decoder.mid_block.resnets.0", top, top)
        a = "decoder.mid_block.attentions.0."
        vec(a + "group_norm", top)
        for s in ("to_q", "to_k", "to_v", "to_out.0"):
            S[a + s + ".weight"], S[a + s + ".bias"] = (top, top), (top,)
        resnet("decoder.mid_block.resnets.1", top, top)
        prev = top
        for i, co in enumerate(reversed(boc)):
            for j in range(c.layers_per_block + 1):
                resnet(f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else co, co)
            prev = co
            if i != len(boc) - 1:
                conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", co, co, 3)
        vec("decoder.conv_norm_out", boc[0])
        conv("decoder.conv_out", c.out_channels, boc[0], 3)
        return S

    def encoder_param_shapes(self):
        """{state-dict key: shape} of what encode() needs (encoder.* + quant_conv.*)."""
        c = self.config
        boc, top, lc = c.block_out_channels, c.block_out_channels[-1], c.latent_channels
        S = {}

        def conv(n, co, ci, k):
            S[n + ".weight"], S[n + ".bias"] = (co, ci, k, k), (co,)

        def vec(n, ch):
            S[n + ".weight"], S[n + ".bias"] = (ch,), (ch,)

        def resnet(n, ci, co):
            vec(n + ".norm1", ci)
            conv(n + ".conv1", co, ci, 3)
            vec(n + ".norm2", co)
            conv(n + ".conv2", co, co, 3)
            if ci != co:
                conv(n + ".conv_shortcut", co, ci, 1)
        conv("encoder.conv_in", boc[0], c.in_channels, 3)
        prev = boc[0]
        for i, co in enumerate(boc):
            for j in range(c.layers_per_block):
                resnet(f"encoder.down_blocks.{i}.resnets.{j}", prev if j == 0 else co, co)
            prev = co
            if i != len(boc) - 1:
                conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", co, co, 3)
        resnet("encoder.mid_block.resnets.0", top, top)
        a = "encoder.mid_block.attentions.0."
        vec(a + "group_norm", top)
        for s_ in ("to_q", "to_k", "to_v", "to_out.0"):
            S[a + s_ + ".weight"], S[a + s_ + ".bias"] = (top, top), (top,)
        resnet("encoder.mid_block.resnets.1", top, top)
        vec("encoder.conv_norm_out", top)
        conv("encoder.conv_out", 2 * lc, top, 3)
        conv("quant_conv", 2 * lc, 2 * lc, 1)
        return S

    def expected_keys(self):
        return list(self.param_shapes())

    def load_state_dict(self, sd, strict=True):
        """Decoder + post_quant_conv keys are required (strict); encoder.* / quant_conv.* keys are accepted and ignored.
        Pre-0.19 diffusers checkpoints name the attention projections query/key/value/proj_attn: mapped here."""
        old = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}
        sd = dict(sd)
        for a in ("decoder.mid_block.attentions.0.", "encoder.mid_block.attentions.0."):
            for o, n in old.items():
                for s in (".weight", ".bias"):
                    if a + o + s in sd and a + n + s not in sd:
                        t = sd.pop(a + o + s)
                        sd[a + n + s] = t.reshape(t.shape[0], -1) if s == ".weight" else t
        missing = [k for k in self.expected_keys() if k not in sd]
        if missing and strict:
            raise KeyError(f"AutoencoderKL: missing keys {missing[:6]} (+{max(0, len(missing) - 6)})")
        enc = [k for k in self.encoder_param_shapes() if k not in sd]
        self.has_encoder = not enc
        if enc and any(k.startswith("encoder.") for k in sd) and strict:   # a partial encoder is a broken checkpoint
            raise KeyError(f"AutoencoderKL: incomplete encoder, missing {enc[:6]} (+{max(0, len(enc) - 6)})")
        self._sd, self._P = sd, None
        return missing, []

    def to(self, device=None, dtype=None, precision=None):
        """dtype fp16 / bf16 / fp32 as in the reference (fp32 = its upcast_vae()); ``precision`` overrides the operand mode."""
        old = (self.device, self.operand_dtype, self.split)
        if device is not None:
            self.device = torch.device(device)
        if dtype is not None:
            assert dtype in (torch.float16, torch.bfloat16, torch.float32)
            self.dtype = dtype
        if precision is not None:
            assert precision in ("auto", "fast", "fp32")
            self.precision = precision
        if (self.device, self.operand_dtype, self.split) != old:
            if self._P is not None and self._sd is None:
                raise RuntimeError("weights were already packed for %s/%s (split=%s); reload the state dict to change them" % old)
            self._P = None
        return self

    def eval(self):
        return self

    def requires_grad_(self, flag=False):
        return self

    # ---- packing ---------------------------------------------------------------------------------------------------------
    def _pack(self):
        if self._P is not None:
            return self._P
        if self._sd is None:
            raise RuntimeError("AutoencoderKL: load_state_dict() first")
        if self.device is None or self.device.type != "cuda":
            raise RuntimeError("AutoencoderKL runs on the GPU only")
        sd, dev, dt, c = self._sd, self.device, self.operand_dtype, self.config
        split = self.split

        rounded = self.dtype == torch.float16      # `.to(dtype=torch.float16)` rounds every parameter; the reference's upcast_vae widens THOSE

        def f32(k):
            t = sd[k].detach().to(dev, torch.float32)
            return (t.to(torch.float16).float() if rounded else t).contiguous()

        two = split and rounded and os.environ.get("SX_VAE_F16X2", "1") != "0"     # two fp16 activation planes x one exact fp16 weight plane

        def conv16x2(k):
            """[Co, Ci, 3, 3] exact-in-fp16 weights → fp16 [Co, 9 * 2Ci]: per tap [W | W] against activations [hi | lo]."""
            w = conv32(k).reshape(-1, 9, sd[k].shape[1]).to(torch.float16)
            return torch.cat([w, w], dim=2).reshape(w.shape[0], -1).contiguous()

        def opnd(w32, taps=1, role="w"):
            """fp32 [N, taps*C] on the device → MFMA operand rows: 16-bit as is, or (fp32-grade mode) bf16 [N, taps*3C] with
            the planes [hi | lo | hi] per tap (role "w"; "a" = [hi | hi | lo] for a weight used as the GEMM's A operand)."""
            return pack_planes(w32, taps, role) if split else w32.to(dt).contiguous()

        def lin16(k, role="w"):
            w = f32(k)
            return opnd(w.reshape(w.shape[0], -1), role=role)

        def conv32(k):  # [Co,Ci,3,3] → fp32 [Co, 9*Ci] with (ky,kx,ci) order
            w = f32(k)
            return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)

        def conv16(k):
            return opnd(conv32(k), taps=9)

        def padded(w32, rows, cols, taps=1):
            wp = torch.zeros(rows, cols, dtype=torch.float32, device=dev)
            wp[:w32.shape[0], :w32.shape[1]] = w32
            return opnd(wp, taps=taps)

        def resnet(n):
            short = n + ".conv_shortcut.weight" in sd
            # p1 / p2: operand planes of conv1 / conv2 (3 = bf16 [hi | hi | lo], 2 = fp16 [hi | lo], 1 = plain). A block with a 1x1
            # shortcut hands norm1's RAW input to it in the same format as the normalised one: that block's conv1 stays at 3 planes
            p1 = (2 if (two and not short) else 3) if split else 1
            p2 = (2 if two else 3) if split else 1
            r = dict(n1=(f32(n + ".norm1.weight"), f32(n + ".norm1.bias")),
                     w1=conv16x2(n + ".conv1.weight") if p1 == 2 else conv16(n + ".conv1.weight"),
                     b1=f32(n + ".conv1.bias"), n2=(f32(n + ".norm2.weight"), f32(n + ".norm2.bias")),
                     w2=conv16x2(n + ".conv2.weight") if p2 == 2 else conv16(n + ".conv2.weight"), b2=f32(n + ".conv2.bias"),
                     p1=p1, p2=p2)
            if short:
                r["ws"], r["bs"] = lin16(n + ".conv_shortcut.weight"), f32(n + ".conv_shortcut.bias")
            return r

        lc, boc = c.latent_channels, c.block_out_channels
        assert lc == 4, "post_quant_conv / conv_in packing assumes 4 latent channels"
        top = boc[-1]
        P = {}
        bpq = torch.zeros(16, dtype=torch.float32, device=dev)
        bpq[:lc] = f32("post_quant_conv.bias")
        P["pq"] = (padded(f32("post_quant_conv.weight").reshape(lc, lc), 16, 64), bpq)   # 1x1 conv as a K-padded GEMM
        self.cin_kpad = (9 * lc + 63) // 64 * 64
        P["conv_in"] = (padded(conv32("decoder.conv_in.weight"), top, self.cin_kpad), f32("decoder.conv_in.bias"))
        a = "decoder.mid_block.attentions.0."
        P["mid"] = dict(r0=resnet("decoder.mid_block.resnets.0"), r1=resnet("decoder.mid_block.resnets.1"),
                        gn=(f32(a + "group_norm.weight"), f32(a + "group_norm.bias")),
                        wq=lin16(a + "to_q.weight"), bq=f32(a + "to_q.bias"),
                        wk=lin16(a + "to_k.weight"), bk=f32(a + "to_k.bias"),
                        wv=lin16(a + "to_v.weight", role="a"), bv=f32(a + "to_v.bias"),
                        wo=lin16(a + "to_out.0.weight"), bo=f32(a + "to_out.0.bias"))
        P["up"] = []
        for i in range(len(boc)):
            blk = dict(res=[resnet(f"decoder.up_blocks.{i}.resnets.{j}") for j in range(c.layers_per_block + 1)], up=None)
            if i != len(boc) - 1:
                n = f"decoder.up_blocks.{i}.upsamplers.0.conv"
                blk["up"] = (conv16(n + ".weight"), f32(n + ".bias"))
            P["up"].append(blk)
        P["norm_out"] = (f32("decoder.conv_norm_out.weight"), f32("decoder.conv_norm_out.bias"))
        wo = conv32("decoder.conv_out.weight")
        bo = torch.zeros(16, dtype=torch.float32, device=dev)
        bo[:wo.shape[0]] = f32("decoder.conv_out.bias")
        if two:
            w16 = torch.zeros(16, 9, wo.shape[1] // 9, dtype=torch.float16, device=dev)
            w16[:wo.shape[0]] = wo.reshape(wo.shape[0], 9, -1).to(torch.float16)
            P["conv_out"] = (torch.cat([w16, w16], dim=2).reshape(16, -1).contiguous(), bo)
        else:
            P["conv_out"] = (padded(wo, 16, wo.shape[1], taps=9), bo)
        P["p_out"] = (2 if two else 3) if split else 1
        P["two"] = two
        if self.has_encoder:
            E = {}
            ci = c.in_channels
            self.enc_kpad = (9 * ci + 63) // 64 * 64
            E["conv_in"] = (padded(conv32("encoder.conv_in.weight"), boc[0], self.enc_kpad), f32("encoder.conv_in.bias"))
            E["down"] = []
            for i in range(len(boc)):
                blk = dict(res=[resnet(f"encoder.down_blocks.{i}.resnets.{j}") for j in range(c.layers_per_block)], down=None)
                if i != len(boc) - 1:
                    n = f"encoder.down_blocks.{i}.downsamplers.0.conv"
                    blk["down"] = (conv16(n + ".weight"), f32(n + ".bias"))
                E["down"].append(blk)
            a = "encoder.mid_block.attentions.0."
            E["mid"] = dict(r0=resnet("encoder.mid_block.resnets.0"), r1=resnet("encoder.mid_block.resnets.1"),
                            gn=(f32(a + "group_norm.weight"), f32(a + "group_norm.bias")),
                            wq=lin16(a + "to_q.weight"), bq=f32(a + "to_q.bias"),
                            wk=lin16(a + "to_k.weight"), bk=f32(a + "to_k.bias"),
                            wv=lin16(a + "to_v.weight", role="a"), bv=f32(a + "to_v.bias"),
                            wo=lin16(a + "to_out.0.weight"), bo=f32(a + "to_out.0.bias"))
            E["norm_out"] = (f32("encoder.conv_norm_out.weight"), f32("encoder.conv_norm_out.bias"))
            wo = conv32("encoder.conv_out.weight")                                  # [2*lc, 9*top]
            bo = torch.zeros(16, dtype=torch.float32, device=dev)
            bo[:wo.shape[0]] = f32("encoder.conv_out.bias")
            E["conv_out"] = (padded(wo, 16, wo.shape[1], taps=9), bo)
            bq = torch.zeros(16, dtype=torch.float32, device=dev)
            bq[:2 * lc] = f32("quant_conv.bias")
            E["quant"] = (padded(f32("quant_conv.weight").reshape(2 * lc, 2 * lc), 16, 64), bq)   # 1x1 as a K-padded GEMM
            P["enc"] = E
        self._P, self._sd = P, None
        return P

    # ---- building blocks -------------------------------------------------------------------------------------------------
    # Every GEMM / conv below is ops.gemm / ops.conv3x3 with fp32 output; in the fp32-grade mode its operands simply carry
    # three bf16 planes per channel (K is tripled), see ops.split_bf16.
    def _A(self, x32, role="a"):
        """fp32 activation [..., C] → MFMA operand ([..., C] 16-bit, or [..., 3C] bf16 planes)."""
        return ops.split_bf16(x32, role) if self.split else ops.cast(x32, self.operand_dtype)

    def _gn(self, x, gb, silu, want_raw=False, stats=None, planes=3):
        """GroupNorm(+SiLU) of the fp32 stream → operand (and optionally the un-normalised x as an operand).
        stats: ops.GnStats accumulated by the conv that produced x (its statistics pass is then skipped).
        planes (fp32-grade mode): 3 = bf16 [hi | hi | lo], 2 = fp16 [hi | lo] (the consumer's weights are one exact fp16 plane)."""
        G = self.config.norm_num_groups
        if not self.split:
            return ops.groupnorm(x, gb[0], gb[1], G, EPS, silu, self.operand_dtype, want_raw=want_raw, stats=stats)
        return ops.groupnorm(x, gb[0], gb[1], G, EPS, silu, None, want_raw=want_raw, planes=planes, stats=stats)

    def _conv(self, planes, *a, **kw):
        """ops.conv3x3 with the bench's FLOP accounting told how many operand planes this launch carries."""
        old, ops.OPERAND_PLANES = ops.OPERAND_PLANES, planes
        try:
            return ops.conv3x3(*a, **kw)
        finally:
            ops.OPERAND_PLANES = old

    def _resnet(self, r, x, H, W):
        """x: fp32 [1, HW, Ci] → fp32 [1, HW, Co]  (ResnetBlock2D with temb=None [ext])."""
        f32 = torch.float32
        Ci = x.shape[-1]
        if "ws" in r:
            h, raw = self._gn(x, r["n1"], True, want_raw=True, planes=r["p1"])
        else:
            h = self._gn(x, r["n1"], True, planes=r["p1"])
        # norm2's statistics ride on conv1's epilogue where conv1 runs on a ping-pong tile (Cout >= 256; ops.GnStats)
        st = None
        if r["w1"].shape[0] >= 256 and (H * W) % 256 == 0:
            st = ops.GnStats(torch.zeros((1, self.config.norm_num_groups, 2), dtype=torch.float64, device=x.device),
                             self.config.norm_num_groups, H * W)
        h = self._conv(r["p1"], h.view(1, H, W, -1), r["w1"], bias=r["b1"], out_dtype=f32, gn=st)
        h = self._gn(h, r["n2"], True, stats=st, planes=r["p2"])
        sc = ops.gemm(raw.view(H * W, -1), r["ws"], bias=r["bs"], out_dtype=f32) if "ws" in r else x.view(-1, Ci)
        return self._conv(r["p2"], h.view(1, H, W, -1), r["w2"], bias=r["b2"], residual=sc, out_dtype=f32)

    def _mid_attention(self, m, x, q_chunk=2048):
        """Attention(heads=1, dim_head=C, residual_connection=True) over the HW pixels of ONE image. x: fp32 [1, HW, C].
        The single head is C = 512 wide — beyond the flash kernel's 128 — so scores go through the GEMM kernel, but in
        blocks of ``q_chunk`` query rows: the fp32 score block is q_chunk x HW (128 MB at 16 384 pixels) instead of the
        1-GiB full matrix, and each block's softmax / P·V run while it is still cache-warm."""
        dt, split, f32 = self.operand_dtype, self.split, torch.float32
        _, HW, C = x.shape
        assert HW % 64 == 0, "latent H·W must be a multiple of 64 (it is the K of the P·V GEMM)"
        G = self.config.norm_num_groups
        if split:
            t32 = ops.groupnorm(x, m["gn"][0], m["gn"][1], G, EPS, False, f32).view(HW, C)
            t = ops.split_bf16(t32)
            q = ops.split_bf16(ops.gemm(t, m["wq"], bias=m["bq"], out_dtype=f32))
            k = ops.split_bf16(ops.gemm(t, m["wk"], bias=m["bk"], out_dtype=f32), "w")
            vt = ops.split_bf16(ops.gemm(m["wv"], ops.split_bf16(t32, "w"), out_dtype=f32), "w")   # [C, 3·HW]
            o = torch.empty((HW, C), dtype=f32, device=x.device)
        else:
            t = ops.groupnorm(x, m["gn"][0], m["gn"][1], G, EPS, False, dt).view(HW, C)
            q, k = ops.gemm(t, m["wq"], bias=m["bq"]), ops.gemm(t, m["wk"], bias=m["bk"])
            vt = ops.gemm(m["wv"], t)                                                # [C, HW] = W_v · X^T = (X · W_v^T)^T
            o = torch.empty((HW, C), dtype=dt, device=x.device)
        scale = 1.0 / math.sqrt(C)
        for q0 in range(0, HW, q_chunk):
            q1 = min(HW, q0 + q_chunk)
            scores = ops.gemm(q[q0:q1], k, out_dtype=f32)                            # [rows, HW] = q · k^T
            p = ops.softmax_rows(scores, scale, f32 if split else dt)
            ops.gemm(ops.split_bf16(p) if split else p, vt, bias=m["bv"], out=o[q0:q1],
                     out_dtype=o.dtype)                                              # P · V + b_v  (rows of P sum to 1)
        o = ops.split_bf16(o) if split else o
        return ops.gemm(o, m["wo"], bias=m["bo"], residual=x.view(HW, C), out_dtype=f32).view(1, HW, C)

    def _im2col(self, x, kpad):
        """3x3 patches of a few-channel fp32 map [1, H, W, c] as the K-padded operand of the stem conv. A pure gather, so in
        the fp32-grade mode the hi and lo planes are gathered separately (exact) and laid out [hi | hi | lo]."""
        if not self.split:
            return ops.im2col3x3_small(x, kpad, self.operand_dtype)
        pl = ops.split_bf16(x.view(-1, 4))                                           # [n/4, 12] = [hi | hi | lo] per 4 values
        hi, lo = (ops.im2col3x3_small(ops.cast(pl[:, a:a + 4].contiguous(), torch.float32).view(x.shape), kpad,
                                      torch.bfloat16) for a in (0, 8))
        return torch.cat([hi, hi, lo], dim=1)

    def _decode_one(self, z_nchw):
        """z: fp32 [1, latent, h, w] → fp32 [1, 3, 8h, 8w] (for the 4-level SDXL config)."""
        P, c, f32 = self._P, self.config, torch.float32
        _, lc, h, w = z_nchw.shape
        zp = ops.nchw_to_nhwc(z_nchw, ld=64)                                         # [1, hw, 64] fp32, channels ≥ lc are 0
        x = ops.gemm(self._A(zp.view(-1, 64)), P["pq"][0], bias=P["pq"][1], out_dtype=f32, n_valid=4)
        col = self._im2col(x.view(1, h, w, lc), self.cin_kpad)
        x = ops.gemm(col, P["conv_in"][0], bias=P["conv_in"][1], out_dtype=f32).view(1, h * w, -1)
        x = self._resnet(P["mid"]["r0"], x, h, w)
        x = self._mid_attention(P["mid"], x)
        x = self._resnet(P["mid"]["r1"], x, h, w)
        H, W = h, w
        for blk in P["up"]:
            for r in blk["res"]:
                x = self._resnet(r, x, H, W)
            if blk["up"] is not None:
                x = ops.conv3x3(self._A(x).view(1, H, W, -1), blk["up"][0], bias=blk["up"][1], upsample=True, out_dtype=f32)
                H, W = 2 * H, 2 * W
        hN = self._gn(x, P["norm_out"], True, planes=P["p_out"])
        y = self._conv(P["p_out"], hN.view(1, H, W, -1), P["conv_out"][0], bias=P["conv_out"][1], out_dtype=f32,
                       n_valid=4)                                                    # [1, HW, 4]: RGB + one zero column
        return ops.nhwc_to_nchw(y, c.out_channels, H, W)

    @torch.no_grad()
    def decode(self, z, return_dict=True, generator=None):
        """z: [B, latent_channels, h, w] latents ALREADY divided by config.scaling_factor (the call site does that).
        Returns DecoderOutput(sample=[B, 3, 8h, 8w] fp32) or a 1-tuple."""
        self._pack()
        z = z.to(device=self.device, dtype=torch.float32)
        ops.OPERAND_PLANES = 3 if self.split else 1
        try:
            out = torch.cat([self._decode_one(z[b:b + 1].contiguous()) for b in range(z.shape[0])], dim=0)
        finally:
            ops.OPERAND_PLANES = 1
        return DecoderOutput(out) if return_dict else (out,)

    def _encode_one(self, img_nchw):
        """img: fp32 [1, 3, H, W] in [-1, 1] → mean of the posterior, fp32 [1, latent, H/8, W/8]."""
        P, c, f32 = self._P, self.config, torch.float32
        E = P["enc"]
        _, ci, H, W = img_nchw.shape
        nb = len(c.block_out_channels)
        assert H % (1 << (nb - 1)) == 0 and W % (1 << (nb - 1)) == 0
        x = ops.nchw_to_nhwc(img_nchw)                                               # [1, HW, 3] fp32
        col = self._im2col(x.view(1, H, W, ci), self.enc_kpad)
        x = ops.gemm(col, E["conv_in"][0], bias=E["conv_in"][1], out_dtype=f32).view(1, H * W, -1)
        for blk in E["down"]:
            for r in blk["res"]:
                x = self._resnet(r, x, H, W)
            if blk["down"] is not None:                                              # Downsample2D(padding=0) + F.pad(0,1,0,1)
                x = ops.conv3x3(self._A(x).view(1, H, W, -1), blk["down"][0], bias=blk["down"][1], stride=2, pad_mode=1,
                                out_dtype=f32)
                H, W = H // 2, W // 2
        x = self._resnet(E["mid"]["r0"], x, H, W)
        x = self._mid_attention(E["mid"], x)
        x = self._resnet(E["mid"]["r1"], x, H, W)
        hN = self._gn(x, E["norm_out"], True)
        lc2 = 2 * c.latent_channels
        y = ops.conv3x3(hN.view(1, H, W, -1), E["conv_out"][0], bias=E["conv_out"][1], out_dtype=f32,
                        n_valid=lc2)                                                 # [1, HW, 8] moments before quant_conv
        pad = torch.zeros((H * W, 64), dtype=torch.float32, device=y.device)
        ops.copy2d(y.view(H * W, lc2), pad, 0)
        m = ops.gemm(self._A(pad), E["quant"][0], bias=E["quant"][1], out_dtype=f32, n_valid=lc2)
        return ops.nhwc_to_nchw(m.view(1, H * W, lc2), c.latent_channels, H, W)     # mean = first latent_channels columns

    @torch.no_grad()
    def encode(self, x, return_dict=True):
        """x: [B, 3, H, W] in [-1, 1]. Returns an object with ``.latent_dist.mode()`` (the reference's only use, :520-523)."""
        if not self.has_encoder:
            raise NotImplementedError("this AutoencoderKL was loaded without encoder.* / quant_conv.* weights")
        self._pack()
        x = x.to(device=self.device, dtype=torch.float32)
        ops.OPERAND_PLANES = 3 if self.split else 1
        try:
            mean = torch.cat([self._encode_one(x[b:b + 1].contiguous()) for b in range(x.shape[0])], dim=0)
        finally:
            ops.OPERAND_PLANES = 1
        dist = SimpleNamespace(mode=lambda: mean, mean=mean)
        out = SimpleNamespace(latent_dist=dist)
        return out if return_dict else (dist,)
