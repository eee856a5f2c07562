"""SDXL VAE decoder on the HIP kernels (the last stage of "image out").

Stand-in for ``diffusers.AutoencoderKL`` [ext, diffusers 0.25.0] as the reference uses it:
``image = vae.decode(latents / vae.config.scaling_factor, return_dict=False)[0]``
(pipeline_stable_diffusion_xl_t2i_edit.py:965-977, after ``upcast_vae`` :569-586 because the fp16 VAE overflows). Same
constructor config names, ``from_pretrained`` directory layout (``config.json`` + ``diffusion_pytorch_model.*``), state-dict
keys, ``.config.scaling_factor`` / ``.config.force_upcast`` / ``.dtype`` and ``decode()`` signature.

Design: NHWC end to end, fp32 residual stream with 16-bit MFMA operands (bf16 has fp32's exponent range, so the fp16
overflow that forces the reference into fp32 does not arise; fp16 operands are accepted but inherit that risk with real
weights). 3×3 convolutions = the implicit-GEMM kernel (nearest-2× upsample fused into the conv's gather), GroupNorm+SiLU
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
        self._sd, self._P = None, None
        self.has_encoder = False

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

    def to(self, device=None, dtype=None):
        old = (self.device, self.dtype)
        if device is not None:
            self.device = torch.device(device)
        if dtype is not None:
            if dtype == torch.float32:          # the reference's upcast_vae(): our accumulation/residual path already is fp32
                dtype = torch.bfloat16
            assert dtype in (torch.float16, torch.bfloat16)
            self.dtype = dtype
        if (self.device, self.dtype) != old:
            if self._P is not None and self._sd is None:
                raise RuntimeError("weights were already packed for %s/%s; reload the state dict to move them" % old)
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
        sd, dev, dt, c = self._sd, self.device, self.dtype, self.config

        def f32(k):
            return sd[k].detach().to(dev, torch.float32).contiguous()

        def lin16(k):
            w = sd[k].detach().to(dev, dt)
            return w.reshape(w.shape[0], -1).contiguous()

        def conv16(k):  # [Co,Ci,3,3] → [Co, 9*Ci] with (ky,kx,ci) order
            w = sd[k].detach().to(dev, dt)
            return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()

        def resnet(n):
            r = dict(n1=(f32(n + ".norm1.weight"), f32(n + ".norm1.bias")), w1=conv16(n + ".conv1.weight"),
                     b1=f32(n + ".conv1.bias"), n2=(f32(n + ".norm2.weight"), f32(n + ".norm2.bias")),
                     w2=conv16(n + ".conv2.weight"), b2=f32(n + ".conv2.bias"))
            if n + ".conv_shortcut.weight" in sd:
                r["ws"], r["bs"] = lin16(n + ".conv_shortcut.weight"), f32(n + ".conv_shortcut.bias")
            return r

        lc, boc = c.latent_channels, c.block_out_channels
        assert lc == 4, "post_quant_conv / conv_in packing assumes 4 latent channels"
        top = boc[-1]
        P = {}
        wpq = torch.zeros(16, 64, dtype=dt, device=dev)                     # 1x1 conv as a K-padded GEMM
        wpq[:lc, :lc] = sd["post_quant_conv.weight"].detach().to(dev, dt).reshape(lc, lc)
        bpq = torch.zeros(16, dtype=torch.float32, device=dev)
        bpq[:lc] = f32("post_quant_conv.bias")
        P["pq"] = (wpq, bpq)
        self.cin_kpad = (9 * lc + 63) // 64 * 64
        w = sd["decoder.conv_in.weight"].detach().to(dev, dt).permute(0, 2, 3, 1).reshape(top, -1)
        wp = torch.zeros(top, self.cin_kpad, dtype=dt, device=dev)
        wp[:, :w.shape[1]] = w
        P["conv_in"] = (wp, f32("decoder.conv_in.bias"))
        a = "decoder.mid_block.attentions.0."
        P["mid"] = dict(r0=resnet("decoder.mid_block.resnets.0"), r1=resnet("decoder.mid_block.resnets.1"),
                        gn=(f32(a + "group_norm.weight"), f32(a + "group_norm.bias")),
                        wq=lin16(a + "to_q.weight"), bq=f32(a + "to_q.bias"),
                        wk=lin16(a + "to_k.weight"), bk=f32(a + "to_k.bias"),
                        wv=lin16(a + "to_v.weight"), bv=f32(a + "to_v.bias"),
                        wo=lin16(a + "to_out.0.weight"), bo=f32(a + "to_out.0.bias"))
        P["up"] = []
        for i in range(len(boc)):
            blk = dict(res=[resnet(f"decoder.up_blocks.{i}.resnets.{j}") for j in range(c.layers_per_block + 1)], up=None)
            if i != len(boc) - 1:
                n = f"decoder.up_blocks.{i}.upsamplers.0.conv"
                blk["up"] = (conv16(n + ".weight"), f32(n + ".bias"))
            P["up"].append(blk)
        P["norm_out"] = (f32("decoder.conv_norm_out.weight"), f32("decoder.conv_norm_out.bias"))
        wo = conv16("decoder.conv_out.weight")
        wop = torch.zeros(16, wo.shape[1], dtype=dt, device=dev)
        wop[:wo.shape[0]] = wo
        bo = torch.zeros(16, dtype=torch.float32, device=dev)
        bo[:wo.shape[0]] = f32("decoder.conv_out.bias")
        P["conv_out"] = (wop, bo)
        if self.has_encoder:
            E = {}
            ci = c.in_channels
            self.enc_kpad = (9 * ci + 63) // 64 * 64
            w = sd["encoder.conv_in.weight"].detach().to(dev, dt).permute(0, 2, 3, 1).reshape(boc[0], -1)
            wp = torch.zeros(boc[0], self.enc_kpad, dtype=dt, device=dev)
            wp[:, :w.shape[1]] = w
            E["conv_in"] = (wp, f32("encoder.conv_in.bias"))
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
                            wv=lin16(a + "to_v.weight"), bv=f32(a + "to_v.bias"),
                            wo=lin16(a + "to_out.0.weight"), bo=f32(a + "to_out.0.bias"))
            E["norm_out"] = (f32("encoder.conv_norm_out.weight"), f32("encoder.conv_norm_out.bias"))
            wo = conv16("encoder.conv_out.weight")                                  # [2*lc, 9*top]
            wop = torch.zeros(16, wo.shape[1], dtype=dt, device=dev)
            wop[:wo.shape[0]] = wo
            bo = torch.zeros(16, dtype=torch.float32, device=dev)
            bo[:wo.shape[0]] = f32("encoder.conv_out.bias")
            E["conv_out"] = (wop, bo)
            wq = torch.zeros(16, 64, dtype=dt, device=dev)                          # quant_conv 1x1 as a K-padded GEMM
            wq[:2 * lc, :2 * lc] = sd["quant_conv.weight"].detach().to(dev, dt).reshape(2 * lc, 2 * lc)
            bq = torch.zeros(16, dtype=torch.float32, device=dev)
            bq[:2 * lc] = f32("quant_conv.bias")
            E["quant"] = (wq, bq)
            P["enc"] = E
        self._P, self._sd = P, None
        return P

    # ---- building blocks -------------------------------------------------------------------------------------------------
    def _resnet(self, r, x, H, W):
        """x: fp32 [1, HW, Ci] → fp32 [1, HW, Co]  (ResnetBlock2D with temb=None [ext])."""
        G, dt = self.config.norm_num_groups, self.dtype
        Ci = x.shape[-1]
        if "ws" in r:
            h, raw = ops.groupnorm(x, r["n1"][0], r["n1"][1], G, EPS, True, dt, want_raw=True)
        else:
            h = ops.groupnorm(x, r["n1"][0], r["n1"][1], G, EPS, True, dt)
        h = ops.conv3x3(h.view(1, H, W, Ci), r["w1"], bias=r["b1"], out_dtype=torch.float32)
        Co = h.shape[-1]
        h = ops.groupnorm(h, r["n2"][0], r["n2"][1], G, EPS, True, dt)
        sc = ops.gemm(raw.view(-1, Ci), r["ws"], bias=r["bs"], out_dtype=torch.float32) if "ws" in r else x.view(-1, Ci)
        return ops.conv3x3(h.view(1, H, W, Co), r["w2"], bias=r["b2"], residual=sc, out_dtype=torch.float32)

    def _mid_attention(self, m, x, q_chunk=2048):
        """Attention(heads=1, dim_head=C, residual_connection=True) over the HW pixels of ONE image. x: fp32 [1, HW, C].
        The single head is C = 512 wide — beyond the flash kernel's 128 — so scores go through the GEMM kernel, but in
        blocks of ``q_chunk`` query rows: the fp32 score block is q_chunk x HW (128 MB at 16 384 pixels) instead of the
        1-GiB full matrix, and each block's softmax / P·V run while it is still cache-warm."""
        dt = self.dtype
        _, HW, C = x.shape
        assert HW % 64 == 0, "latent H·W must be a multiple of 64 (it is the K of the P·V GEMM)"
        t = ops.groupnorm(x, m["gn"][0], m["gn"][1], self.config.norm_num_groups, EPS, False, dt).view(HW, C)
        q, k = ops.gemm(t, m["wq"], bias=m["bq"]), ops.gemm(t, m["wk"], bias=m["bk"])
        vt = ops.gemm(m["wv"], t)                                                    # [C, HW] = W_v · X^T = (X · W_v^T)^T
        o = torch.empty((HW, C), dtype=dt, device=x.device)
        scale = 1.0 / math.sqrt(C)
        for q0 in range(0, HW, q_chunk):
            q1 = min(HW, q0 + q_chunk)
            scores = ops.gemm(q[q0:q1], k, out_dtype=torch.float32)                  # [rows, HW] = q · k^T
            p = ops.softmax_rows(scores, scale, dt)
            ops.gemm(p, vt, bias=m["bv"], out=o[q0:q1])                              # P · V + b_v  (rows of P sum to 1)
        return ops.gemm(o, m["wo"], bias=m["bo"], residual=x.view(HW, C), out_dtype=torch.float32).view(1, HW, C)

    def _decode_one(self, z_nchw):
        """z: fp32 [1, latent, h, w] → fp32 [1, 3, 8h, 8w] (for the 4-level SDXL config)."""
        P, dt, c = self._P, self.dtype, self.config
        _, lc, h, w = z_nchw.shape
        zp = ops.nchw_to_nhwc(z_nchw, ld=64)                                         # [1, hw, 64] fp32, channels ≥ lc are 0
        x = ops.gemm(ops.cast(zp.view(-1, 64), dt), P["pq"][0], bias=P["pq"][1], out_dtype=torch.float32, n_valid=4)
        col = ops.im2col3x3_small(x.view(1, h, w, lc), self.cin_kpad, dt)
        x = ops.gemm(col, P["conv_in"][0], bias=P["conv_in"][1], out_dtype=torch.float32).view(1, h * w, -1)
        x = self._resnet(P["mid"]["r0"], x, h, w)
        x = self._mid_attention(P["mid"], x)
        x = self._resnet(P["mid"]["r1"], x, h, w)
        H, W = h, w
        for blk in P["up"]:
            for r in blk["res"]:
                x = self._resnet(r, x, H, W)
            if blk["up"] is not None:
                Ci = x.shape[-1]
                x = ops.conv3x3(ops.cast(x, dt).view(1, H, W, Ci), blk["up"][0], bias=blk["up"][1], upsample=True,
                                out_dtype=torch.float32)
                H, W = 2 * H, 2 * W
        hN = ops.groupnorm(x, P["norm_out"][0], P["norm_out"][1], c.norm_num_groups, EPS, True, dt)
        y = ops.conv3x3(hN.view(1, H, W, x.shape[-1]), P["conv_out"][0], bias=P["conv_out"][1], out_dtype=torch.float32,
                        n_valid=4)                                                   # [1, HW, 4]: RGB + one zero column
        return ops.nhwc_to_nchw(y, c.out_channels, H, W)

    @torch.no_grad()
    def decode(self, z, return_dict=True, generator=None):
        """z: [B, latent_channels, h, w] latents ALREADY divided by config.scaling_factor (the call site does that).
        Returns DecoderOutput(sample=[B, 3, 8h, 8w] fp32) or a 1-tuple."""
        self._pack()
        z = z.to(device=self.device, dtype=torch.float32)
        out = torch.cat([self._decode_one(z[b:b + 1].contiguous()) for b in range(z.shape[0])], dim=0)
        return DecoderOutput(out) if return_dict else (out,)

    def _encode_one(self, img_nchw):
        """img: fp32 [1, 3, H, W] in [-1, 1] → mean of the posterior, fp32 [1, latent, H/8, W/8]."""
        P, dt, c = self._P, self.dtype, self.config
        E = P["enc"]
        _, ci, H, W = img_nchw.shape
        nb = len(c.block_out_channels)
        assert H % (1 << (nb - 1)) == 0 and W % (1 << (nb - 1)) == 0
        x = ops.nchw_to_nhwc(img_nchw)                                               # [1, HW, 3] fp32
        col = ops.im2col3x3_small(x.view(1, H, W, ci), self.enc_kpad, dt)
        x = ops.gemm(col, E["conv_in"][0], bias=E["conv_in"][1], out_dtype=torch.float32).view(1, H * W, -1)
        for blk in E["down"]:
            for r in blk["res"]:
                x = self._resnet(r, x, H, W)
            if blk["down"] is not None:                                              # Downsample2D(padding=0) + F.pad(0,1,0,1)
                Ci = x.shape[-1]
                x = ops.conv3x3(ops.cast(x, dt).view(1, H, W, Ci), blk["down"][0], bias=blk["down"][1], stride=2,
                                pad_mode=1, out_dtype=torch.float32)
                H, W = H // 2, W // 2
        x = self._resnet(E["mid"]["r0"], x, H, W)
        x = self._mid_attention(E["mid"], x)
        x = self._resnet(E["mid"]["r1"], x, H, W)
        hN = ops.groupnorm(x, E["norm_out"][0], E["norm_out"][1], c.norm_num_groups, EPS, True, dt)
        lc2 = 2 * c.latent_channels
        y = ops.conv3x3(hN.view(1, H, W, x.shape[-1]), E["conv_out"][0], bias=E["conv_out"][1], out_dtype=torch.float32,
                        n_valid=lc2)                                                 # [1, HW, 8] moments before quant_conv
        pad = torch.zeros((H * W, 64), dtype=torch.float32, device=y.device)
        ops.copy2d(y.view(H * W, lc2), pad, 0)
        m = ops.gemm(ops.cast(pad, dt), E["quant"][0], bias=E["quant"][1], out_dtype=torch.float32, n_valid=lc2)
        return ops.nhwc_to_nchw(m.view(1, H * W, lc2), c.latent_channels, H, W)     # mean = first latent_channels columns

    @torch.no_grad()
    def encode(self, x, return_dict=True):
        """x: [B, 3, H, W] in [-1, 1]. Returns an object with ``.latent_dist.mode()`` (the reference's only use, :520-523)."""
        if not self.has_encoder:
            raise NotImplementedError("this AutoencoderKL was loaded without encoder.* / quant_conv.* weights")
        self._pack()
        x = x.to(device=self.device, dtype=torch.float32)
        mean = torch.cat([self._encode_one(x[b:b + 1].contiguous()) for b in range(x.shape[0])], dim=0)
        dist = SimpleNamespace(mode=lambda: mean, mean=mean)
        out = SimpleNamespace(latent_dist=dist)
        return out if return_dict else (dist,)
