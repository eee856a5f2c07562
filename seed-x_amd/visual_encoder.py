"""Path A — Qwen-VL ViT-bigG visual encoder with attention-pool resampler, on the HIP kernels.

Mirrors the reference interface of ``src/models/tokenizer/qwen_visual.py`` (``VisionTransformerWithAttnPool``:
constructor kwargs :327-337, ``from_pretrained`` :431-459, ``forward(x, patch_positions=None)`` :387-417,
``.eval()`` / ``.to(device, dtype=)``) and accepts the same state-dict key names, so
``configs/visual_encoder/qwen_vitg_448.yaml`` only needs its ``_target_`` repointed.

Execution plan per forward (all dense work in libseedx_hip.so, residual stream fp32, GEMM operands 16-bit):
  patchify → GEMM(+bicubic-resized pos table as broadcast residual) → LN(ln_pre)
  48 × [LN → GEMM(in_proj,+bias) → flash attention on the per-head-interleaved [T,H,3,hd] layout (no permute)
        → GEMM(out_proj,+bias,+residual) → LN → GEMM(c_fc,+bias,GELU) → GEMM(c_proj,+bias,+residual)]
  attn_pool Resampler (constant query branch precomputed at pack time) → LN(ln_post) → GEMM(proj)
"""
import math
import os

import torch
import torch.nn.functional as F

from . import ops


def get_abs_pos(abs_pos, tgt_len):
    """Host-side constant: bicubic resize of a square position table (reference qwen_visual.py:24-40). Done once at
    pack time on the CPU in fp32 — the reference recomputes it every call."""
    src = int(math.sqrt(abs_pos.size(0)))
    tgt = int(math.sqrt(tgt_len))
    if src == tgt:
        return abs_pos.float()
    return F.interpolate(abs_pos.float().reshape(1, src, src, -1).permute(0, 3, 1, 2), size=(tgt, tgt), mode="bicubic",
                         align_corners=False).permute(0, 2, 3, 1).flatten(0, 2)


def get_2d_sincos_pos_embed(embed_dim, grid_size):
    """2-D sin/cos table of the Resampler (reference qwen_visual.py:44-91; w first in the meshgrid)."""
    import numpy as np
    gh = np.arange(grid_size, dtype=np.float32)
    gw = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(gw, gh), axis=0).reshape(2, 1, grid_size, grid_size)

    def one(d, pos):
        omega = np.arange(d // 2, dtype=np.float32)
        omega /= d / 2.0
        omega = 1.0 / 10000 ** omega
        out = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)

    return torch.from_numpy(np.concatenate([one(embed_dim // 2, grid[0]), one(embed_dim // 2, grid[1])], axis=1)).float()


class Resampler:
    """2-D perceiver resampler with one cross-attention layer (reference qwen_visual.py:94-149). Used as the ViT
    attn_pool and as ContinuousLVLM's input/output resamplers (configs/clm_models/agent_seed_x_i.yaml:2-14)."""

    def __init__(self, grid_size, embed_dim, num_heads, kv_dim=None, norm_layer=None, eps=1e-5):
        self.num_queries = grid_size ** 2
        self.grid_size = grid_size
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.kv_dim = kv_dim if kv_dim is not None else embed_dim
        self.out_dim = self.kv_dim if self.kv_dim != embed_dim else embed_dim
        # nn.LayerNorm default eps (1e-5) unless the ViT passes partial(nn.LayerNorm, eps=1e-6) (:358,376)
        self.eps = getattr(norm_layer, "keywords", {}).get("eps", eps) if norm_layer is not None else eps
        self._sd = None
        self._packed = {}
        self.device, self.dtype = None, torch.float16
        # fp32-grade activations (default; SX_RESAMPLER_PRECISE=0 for the plain 16-bit flow): two operand planes into the four GEMMs, fp32 K / V
        # and attention (csrc/precise.hip). The input resampler's output IS the LLM's image-token input and the output resampler's output IS
        # the de-tokenizer's conditioning: at one 16-bit rounding per operand each sat at 6-7e-4 of the fp32 reference on its own.
        self.precise = os.environ.get("SX_RESAMPLER_PRECISE", "1") != "0"

    def parameter_names(self, prefix=""):
        names = ["pos_embed", "query", "attn.in_proj_weight", "attn.in_proj_bias", "attn.out_proj.weight",
                 "attn.out_proj.bias", "ln_q.weight", "ln_q.bias", "ln_kv.weight", "ln_kv.bias"]
        if self.kv_dim != self.embed_dim:
            names.append("kv_proj.weight")
        return [prefix + n for n in names]

    def load_state_dict(self, sd, prefix="", strict=True):
        own = {}
        missing = []
        for n in self.parameter_names():
            if prefix + n in sd:
                own[n] = sd[prefix + n].detach()
            elif n == "pos_embed":
                own[n] = get_2d_sincos_pos_embed(self.embed_dim, self.grid_size)
            else:
                missing.append(prefix + n)
        if missing and strict:
            raise KeyError(f"Resampler.load_state_dict: missing keys {missing}")
        self._sd = own
        self._packed = {}
        return missing

    def to(self, device=None, dtype=None):
        old = (self.device, self.dtype)
        if device is not None:
            self.device = torch.device(device)
        if dtype is not None:
            self.dtype = dtype
        if (self.device, self.dtype) != old:
            self._packed = {}
        return self

    def eval(self):
        return self

    def _pack(self, n_kv):
        key = n_kv
        if key in self._packed:
            return self._packed[key]
        sd, dev, dt, E = self._sd, self.device, self.dtype, self.embed_dim
        f32 = lambda t: t.to(dev, torch.float32).contiguous()
        w16 = lambda t: t.to(dev, dt).contiguous()
        P = {}
        if "kv_proj.weight" in sd:
            P["kv_w"] = w16(sd["kv_proj.weight"])
        P["ln_kv"] = (f32(sd["ln_kv.weight"]), f32(sd["ln_kv.bias"]))
        Win, bin_ = sd["attn.in_proj_weight"], sd["attn.in_proj_bias"].float().cpu()
        # K/V projections fused: [2E, E]; the key-side position term in_proj_k(pos↑)+b_k is a constant table
        P["kv_in_w"] = w16(Win[E:])
        pos_k = get_abs_pos(sd["pos_embed"].float().cpu(), n_kv)                   # [n_kv, E] fp32 (host, constant)
        kb = pos_k @ Win[E:2 * E].float().cpu().t() + bin_[E:2 * E]                # fp32 host precompute (load time)
        vb = bin_[2 * E:].unsqueeze(0).expand(n_kv, E)
        P["kv_res"] = f32(torch.cat([kb, vb], dim=1))                              # residual table, rows m % n_kv
        P["out_w"] = w16(sd["attn.out_proj.weight"])
        P["out_b"] = f32(sd["attn.out_proj.bias"])
        # constant query branch: q = in_proj_q(ln_q(query) + pos_embed) — computed once on the GPU kernels
        q = ops.layernorm(f32(sd["query"]), f32(sd["ln_q.weight"]), f32(sd["ln_q.bias"]), self.eps, torch.float32)
        q = ops.add(q, f32(sd["pos_embed"]))
        P["q"] = ops.gemm(ops.cast(q, dt), w16(Win[:E]), bias=f32(bin_[:E]))       # [nq, E] 16-bit
        P["q32"] = ops.gemm(ops.split16(q, dt), w16(Win[:E]), a_planes=2, bias=f32(bin_[:E]), out_dtype=torch.float32)   # precise mode
        self._packed[key] = P
        return P

    def forward(self, x, attn_mask=None):
        """x: [B, n_kv, kv_dim] (any float dtype, on the GPU) → [B, n_queries, embed_dim] fp32."""
        assert attn_mask is None, "attn_mask is never used on the inference path"
        B, n_kv, _ = x.shape
        P = self._pack(n_kv)
        E, H = self.embed_dim, self.num_heads
        hd = E // H
        if self.precise and hd % 8 == 0 and hd <= 256:
            f32, dt, nq = torch.float32, self.dtype, self.num_queries
            x2 = (x if x.dtype == f32 else ops.cast(x.contiguous(), f32)).reshape(B * n_kv, -1).contiguous()
            if "kv_w" in P:
                x2 = ops.gemm(ops.split16(x2, dt), P["kv_w"], a_planes=2, out_dtype=f32)
            h = ops.layernorm(x2, P["ln_kv"][0], P["ln_kv"][1], self.eps, f32)
            kv = ops.gemm(ops.split16(h, dt), P["kv_in_w"], a_planes=2, residual=P["kv_res"], res_mod=n_kv, out_dtype=f32)   # [B*n_kv, 2E] fp32
            kv5 = kv.view(B, n_kv, 2, H, hd)
            q4 = P["q32"].view(1, nq, H, hd).expand(B, nq, H, hd).contiguous()      # the shared queries, one copy per sample (small)
            att = ops.attention_f32_full(q4, kv5[:, :, 0], kv5[:, :, 1], 1.0 / math.sqrt(hd), dt)            # planes [B*nq, 2E]
            out = ops.gemm(att, P["out_w"], a_planes=2, bias=P["out_b"], out_dtype=f32)
            return out.view(B, nq, E)
        x16 = x if x.dtype == self.dtype else ops.cast(x.contiguous(), self.dtype)
        x2 = x16.reshape(B * n_kv, -1)
        if "kv_w" in P:
            x2 = ops.gemm(x2, P["kv_w"], out_dtype=torch.float32)
        h = ops.layernorm(x2, P["ln_kv"][0], P["ln_kv"][1], self.eps, self.dtype)
        kv = ops.gemm(h, P["kv_in_w"], residual=P["kv_res"], res_mod=n_kv)          # [B*n_kv, 2E]
        kv5 = kv.view(B, n_kv, 2, H, hd)
        nq = self.num_queries
        q4 = P["q"].view(1, nq, H, hd).expand(B, nq, H, hd)                         # batch stride 0: shared queries
        if hd <= 128 and hd % 8 == 0:
            att = ops.attention(q4, kv5[:, :, 0], kv5[:, :, 1], 1.0 / math.sqrt(hd))
        else:
            att = ops.attention_small(q4, kv5[:, :, 0], kv5[:, :, 1], 1.0 / math.sqrt(hd))
        out = ops.gemm(att.view(B * nq, E), P["out_w"], bias=P["out_b"], out_dtype=torch.float32)
        return out.view(B, nq, E)

    __call__ = forward


class VisionTransformerWithAttnPool:
    """Drop-in for reference ``VisionTransformerWithAttnPool`` (qwen_visual.py:325-459)."""

    def __init__(self, image_size, patch_size, width, layers, heads, mlp_ratio, n_queries=256, output_dim=512,
                 patch_pos=False, **kwargs):
        self.image_size, self.patch_size = image_size, patch_size
        self.grid = image_size // patch_size
        self.width, self.layers, self.heads = width, layers, heads
        self.mlp_width = int(width * mlp_ratio)
        self.n_queries, self.output_dim = n_queries, output_dim
        self.patch_pos = patch_pos
        if patch_pos:
            raise NotImplementedError("patch_pos=True is not used by any shipped config (qwen_vitg_448.yaml)")
        self.eps = 1e-6                                                   # partial(nn.LayerNorm, eps=1e-6), :358
        self.attn_pool = Resampler(int(math.sqrt(n_queries)), output_dim, output_dim // 128, kv_dim=width, eps=1e-6)
        self.kpad = (3 * patch_size * patch_size + 63) // 64 * 64
        self._sd = None
        self._P = None
        self.device, self.dtype = None, torch.float16

    # ---- reference-compatible plumbing -----------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, pretrained_model_path=None, **kwargs):
        model = cls(**kwargs)
        if pretrained_model_path is not None:
            ckpt = torch.load(pretrained_model_path, map_location="cpu")   # qwen_vit_G.pt, reference :450-454
            model.load_state_dict(ckpt)
        return model

    def expected_keys(self):
        keys = ["positional_embedding", "proj", "conv1.weight", "ln_pre.weight", "ln_pre.bias", "ln_post.weight",
                "ln_post.bias"]
        for i in range(self.layers):
            p = f"transformer.resblocks.{i}."
            for n in ("ln_1", "ln_2"):
                keys += [p + n + ".weight", p + n + ".bias"]
            for n in ("attn.in_proj", "attn.out_proj", "mlp.c_fc", "mlp.c_proj"):
                keys += [p + n + ".weight", p + n + ".bias"]
        return keys + self.attn_pool.parameter_names("attn_pool.")

    def load_state_dict(self, sd, strict=True):
        """Same key names as the reference checkpoint; unlike the reference's silent non-strict loaders
        (utils.py:7-42) a missing key is an error."""
        missing = [k for k in self.expected_keys() if k not in sd and k != "attn_pool.pos_embed"]
        if missing and strict:
            raise KeyError(f"VisionTransformerWithAttnPool: missing keys {missing[:8]} (+{max(0, len(missing) - 8)})")
        keys = set(self.expected_keys())
        self._sd = {k: v.detach() for k, v in sd.items() if k in keys}
        self.attn_pool.load_state_dict(sd, prefix="attn_pool.", strict=strict)
        self._P = None
        return missing, []

    def to(self, device=None, dtype=None):
        old = (self.device, self.dtype)
        if device is not None:
            self.device = torch.device(device)
        if dtype is not None:
            assert dtype in (torch.float16, torch.bfloat16), "compute dtype must be fp16 or bf16"
            self.dtype = dtype
        self.attn_pool.to(self.device, self.dtype)
        if (self.device, self.dtype) != old:
            if self._P is not None and self._sd is None:
                raise RuntimeError("weights were already packed for %s/%s; reload the state dict to move them" % old)
            self._P = None
        return self

    def eval(self):
        return self

    def get_cast_dtype(self):
        return self.dtype

    # ---- weight packing -------------------------------------------------------------------------------------
    def _pack(self):
        if self._P is not None:
            return self._P
        if self._sd is None:
            raise RuntimeError("VisionTransformerWithAttnPool: load_state_dict() before forward()")
        if self.device is None or self.device.type != "cuda":
            raise RuntimeError("VisionTransformerWithAttnPool runs on the GPU only (call .to('cuda', dtype=...))")
        sd, dev, dt, W = self._sd, self.device, self.dtype, self.width
        f32 = lambda t: t.to(dev, torch.float32).contiguous()
        w16 = lambda t: t.to(dev, dt).contiguous()
        P = {}
        cw = sd["conv1.weight"].reshape(W, -1)
        cwp = torch.zeros(W, self.kpad, dtype=cw.dtype, device=cw.device)
        cwp[:, :cw.shape[1]] = cw
        P["conv_w"] = w16(cwp)
        P["pos"] = f32(get_abs_pos(sd["positional_embedding"].float().cpu(), self.grid * self.grid))
        P["ln_pre"] = (f32(sd["ln_pre.weight"]), f32(sd["ln_pre.bias"]))
        P["layers"] = []
        for i in range(self.layers):
            p = f"transformer.resblocks.{i}."
            P["layers"].append(dict(
                ln1=(f32(sd[p + "ln_1.weight"]), f32(sd[p + "ln_1.bias"])),
                ln2=(f32(sd[p + "ln_2.weight"]), f32(sd[p + "ln_2.bias"])),
                in_w=w16(sd[p + "attn.in_proj.weight"]), in_b=f32(sd[p + "attn.in_proj.bias"]),
                out_w=w16(sd[p + "attn.out_proj.weight"]), out_b=f32(sd[p + "attn.out_proj.bias"]),
                fc_w=w16(sd[p + "mlp.c_fc.weight"]), fc_b=f32(sd[p + "mlp.c_fc.bias"]),
                pj_w=w16(sd[p + "mlp.c_proj.weight"]), pj_b=f32(sd[p + "mlp.c_proj.bias"])))
        P["ln_post"] = (f32(sd["ln_post.weight"]), f32(sd["ln_post.bias"]))
        P["proj_w"] = w16(sd["proj"].t())                                  # x @ proj  ==  x · (proj^T)^T
        self._P = P
        self._sd = None if os.environ.get("SEEDX_KEEP_HOST_WEIGHTS", "0") != "1" else self._sd
        return P

    # ---- forward ----------------------------------------------------------------------------------------------
    def forward(self, x, patch_positions=None):
        """x: [B, 3, S, S] CLIP-normalised image tensor (any float dtype / device) → [B, n_queries, output_dim] in the
        module dtype, exactly like the reference (:387-417)."""
        P = self._pack()
        dt, W, H = self.dtype, self.width, self.heads
        hd = W // H
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        B = x.shape[0]
        L = self.grid * self.grid
        patches = ops.patchify(x, self.patch_size, self.kpad, dt)                              # conv1 as GEMM (:393)
        t = ops.gemm(patches, P["conv_w"], residual=P["pos"], res_mod=L, out_dtype=torch.float32)   # + abs pos (:398)
        xres = ops.layernorm(t, P["ln_pre"][0], P["ln_pre"][1], self.eps, torch.float32)       # ln_pre (:400)
        scale = 1.0 / math.sqrt(hd)                                                            # q / sqrt(hd) (:204)
        for lw in P["layers"]:                                                                 # :312-316
            h = ops.layernorm(xres, lw["ln1"][0], lw["ln1"][1], self.eps, dt)
            qkv = ops.gemm(h, lw["in_w"], bias=lw["in_b"])                                     # [B*L, 3W], [H,3,hd] interleave
            q5 = qkv.view(B, L, H, 3, hd)
            att = ops.attention(q5[:, :, :, 0], q5[:, :, :, 1], q5[:, :, :, 2], scale)         # [B, L, W]
            xres = ops.gemm(att.view(B * L, W), lw["out_w"], bias=lw["out_b"], residual=xres, out_dtype=torch.float32)
            h = ops.layernorm(xres, lw["ln2"][0], lw["ln2"][1], self.eps, dt)
            h = ops.gemm(h, lw["fc_w"], bias=lw["fc_b"], act="gelu")
            xres = ops.gemm(h, lw["pj_w"], bias=lw["pj_b"], residual=xres, out_dtype=torch.float32)
        pooled = self.attn_pool(xres.view(B, L, W))                                            # [B, nq, od] fp32 (:406)
        h = ops.layernorm(pooled.view(-1, self.output_dim), P["ln_post"][0], P["ln_post"][1], self.eps, dt)
        out = ops.gemm(h, P["proj_w"])                                                         # @ proj (:415)
        return out.view(B, self.n_queries, self.output_dim)

    __call__ = forward
