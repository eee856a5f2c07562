"""Path C — SDXL-adapter de-tokenizer on the HIP kernels: ResamplerXLV2, Euler scheduler tables, the two CFG
denoise loops and the SDXLAdapter / SDXLAdapterWithLatentImage front ends.

Reference interfaces mirrored (same names, argument meaning, defaults):
  * ``ResamplerXLV2``                        src/models/detokenizer/resampler.py:226-286
  * ``SDXLAdapter``                          src/models/detokenizer/adapter_modules.py:11-169
  * ``SDXLAdapterWithLatentImage``           adapter_modules.py:172-287
  * t2i loop ``StableDiffusionXLPipeline.__call__`` [ext diffusers 0.25.0] as driven by adapter_modules.py:156-167
  * edit loop ``StableDiffusionXLText2ImageAndEditPipeline.__call__``  pipeline_stable_diffusion_xl_t2i_edit.py:900-963
  * ``EulerDiscreteScheduler`` [ext] with the SDXL scheduler config (eval_seed_x_detokenizer.py:30)
``generate`` returns what the reference returns — a list of PIL images (``sdxl_pipe(...).images``,
adapter_modules.py:156-169 / :273-287) — whenever a VAE was given to ``init_pipe``; without a VAE (the reference would
fail there) it returns the final latents. ``output_type`` ("pil" | "np" | "pt" | "latent" | "raw" | "u8" = device uint8 [B,H,W,3]) overrides, exactly like
the ``**kwargs`` the reference forwards to the diffusers pipeline.

One denoise step = {time embeddings → UNet (CFG batch 2 or 3) → fused CFG + Euler update + next scaled input}, all
device-resident (step counter, sigma table and timestep table live in HBM) and captured once into a HIP graph that is
replayed ``num_inference_steps`` times.
"""
import math
import os

import numpy as np
import torch

from . import ops


# ---------------------------------------------------------------------------------------------------------------
class ResamplerXLV2:
    def __init__(self, dim=1024, depth=8, dim_head=64, heads=16, num_queries=8, embedding_dim=768, output1_dim=768,
                 output2_dim=1280, ff_mult=4, normalize=True):
        self.normalize = normalize                       # F.normalize(x) over the TOKEN axis (resampler.py:271-272)
        self.dim, self.depth, self.dim_head, self.heads = dim, depth, dim_head, heads
        self.num_queries, self.embedding_dim = num_queries, embedding_dim
        self.o1, self.o2, self.ff_mult = output1_dim, output2_dim, ff_mult
        self.in_dim, self.out_dim = dim, output1_dim + output2_dim
        self.device, self.dtype = None, torch.float16
        self._sd, self._P = None, None
        # fp32-grade activations (default): its two outputs condition every UNet step through classifier-free guidance, which multiplies
        # the difference of two conditionings by 7.5 — at one 16-bit rounding per operand (9e-4 / 1.2e-3 on prompt / pooled embeds) the
        # 5-step latents of a full-size generation sat at 1.6e-3 of the fp32 chain on the SAME image features. Operand planes
        # x = hi + lo into sx_gemm a_planes = 2, fp32 q / k / v and attention (csrc/precise.hip); 9 GFLOP per sample, < 0.1 ms more.
        self.precise = os.environ.get("SX_XLV2_PRECISE", "1") != "0"

    def load_state_dict(self, sd, prefix="", strict=True):
        need = ["latents", "proj_in.weight", "norm_out.weight", "unet_proj_1.weight", "unet_attnpool.c_proj.weight"]
        missing = [prefix + k for k in need if prefix + k not in sd]
        if missing and strict:
            raise KeyError(f"ResamplerXLV2: missing keys {missing}")
        self._sd = {k[len(prefix):]: v.detach().float().cpu() for k, v in sd.items() if k.startswith(prefix)}
        self._P = None
        return missing

    def to(self, device=None, dtype=None):
        old = (self.device, self.dtype)
        if device is not None:
            self.device = torch.device(device)
        if dtype is not None:
            self.dtype = dtype
        if (self.device, self.dtype) != old:
            self._P = None
        return self

    def eval(self):
        return self

    def requires_grad_(self, flag=True):
        return self

    def _pack(self):
        if self._P is not None:
            return self._P
        sd, dev, dt = self._sd, self.device, self.dtype
        f32 = lambda k: sd[k].to(dev, torch.float32).contiguous()
        w16 = lambda k: sd[k].to(dev, dt).contiguous()
        P = dict(latents=f32("latents")[0], pin=(w16("proj_in.weight"), f32("proj_in.bias")),
                 nout=(f32("norm_out.weight"), f32("norm_out.bias")), layers=[])
        for i in range(self.depth):
            a, f = f"layers.{i}.0.", f"layers.{i}.1."
            P["layers"].append(dict(n1=(f32(a + "norm1.weight"), f32(a + "norm1.bias")),
                                    n2=(f32(a + "norm2.weight"), f32(a + "norm2.bias")),
                                    wq=w16(a + "to_q.weight"), wkv=w16(a + "to_kv.weight"), wo=w16(a + "to_out.weight"),
                                    fn=(f32(f + "0.weight"), f32(f + "0.bias")), w1=w16(f + "1.weight"),
                                    w3=w16(f + "3.weight")))
        P["proj"] = (torch.cat([sd["unet_proj_1.weight"], sd["unet_proj_2.weight"]], 0).to(dev, dt).contiguous(),
                     torch.cat([sd["unet_proj_1.bias"], sd["unet_proj_2.bias"]], 0).to(dev, torch.float32).contiguous())
        p = "unet_attnpool."
        P["pool_pos"] = f32(p + "positional_embedding")
        P["pool_q"] = (w16(p + "q_proj.weight"), f32(p + "q_proj.bias"))
        P["pool_kv"] = (torch.cat([sd[p + "k_proj.weight"], sd[p + "v_proj.weight"]], 0).to(dev, dt).contiguous(),
                        torch.cat([sd[p + "k_proj.bias"], sd[p + "v_proj.bias"]], 0).to(dev, torch.float32).contiguous())
        P["pool_c"] = (w16(p + "c_proj.weight"), f32(p + "c_proj.bias"))
        self._P = P
        return P

    def forward(self, x, pooled_text_embeds=None):
        """x [B, n, embedding_dim] → (prompt_embeds [B, nq, o1+o2] fp32, pooled [B, o2] fp32)  (resampler.py:266-286)."""
        P, dt, dim, heads, hd = self._pack(), self.dtype, self.dim, self.heads, self.dim_head
        B, n, _ = x.shape
        nq, inner = self.num_queries, self.heads * self.dim_head
        if self.precise:
            return self._forward_precise(x)
        x16 = x.to(self.device)
        if self.normalize:
            from .image_ops import l2norm_dim1
            x16 = l2norm_dim1(x16)
        x16 = x16.contiguous() if x16.dtype == dt else ops.cast(x16.float().contiguous(), dt)
        xs = ops.gemm(x16.view(B * n, -1), P["pin"][0], bias=P["pin"][1], out_dtype=torch.float32)       # proj_in
        lat = P["latents"].unsqueeze(0).expand(B, nq, dim).contiguous().view(B * nq, dim)
        scale = 1.0 / math.sqrt(hd)                     # (q·hd^-¼)·(k·hd^-¼), resampler.py:68-69
        for lw in P["layers"]:
            xn = ops.layernorm(xs, lw["n1"][0], lw["n1"][1], 1e-5, dt)
            ln = ops.layernorm(lat, lw["n2"][0], lw["n2"][1], 1e-5, dt)
            q = ops.gemm(ln, lw["wq"]).view(B, nq, heads, hd)
            kv = torch.empty((B, n + nq, 2 * inner), dtype=dt, device=self.device)                       # cat(x, latents)
            for b in range(B):
                ops.gemm(xn[b * n:(b + 1) * n], lw["wkv"], out=kv[b, :n])
                ops.gemm(ln[b * nq:(b + 1) * nq], lw["wkv"], out=kv[b, n:])
            kv5 = kv.view(B, n + nq, 2, heads, hd)
            att = ops.attention(q, kv5[:, :, 0], kv5[:, :, 1], scale)
            lat = ops.gemm(att.view(B * nq, inner), lw["wo"], residual=lat, out_dtype=torch.float32)
            h = ops.layernorm(lat, lw["fn"][0], lw["fn"][1], 1e-5, dt)
            h = ops.gemm(h, lw["w1"], act="gelu")
            lat = ops.gemm(h, lw["w3"], residual=lat, out_dtype=torch.float32)
        hid = ops.layernorm(lat, P["nout"][0], P["nout"][1], 1e-5, torch.float32)                        # norm_out
        hid16 = ops.cast(hid, dt)
        prompt = ops.gemm(hid16, P["proj"][0], bias=P["proj"][1], out_dtype=torch.float32).view(B, nq, -1)
        # AttentionPool2d (resampler.py:89-116): token 0 = mean token; only its output row is needed
        mean = ops.avgpool_tokens(hid.view(B, nq, dim), nq)                                               # [B,1,dim]
        t = torch.empty((B, nq + 1, dim), dtype=torch.float32, device=self.device)
        t[:, 0:1] = mean
        t[:, 1:] = hid.view(B, nq, dim)
        t = ops.add(t, P["pool_pos"].unsqueeze(0).expand(B, nq + 1, dim).contiguous())
        t16 = ops.cast(t, dt)
        kvp = ops.gemm(t16.view(B * (nq + 1), dim), P["pool_kv"][0], bias=P["pool_kv"][1]).view(B, nq + 1, 2, heads, dim // heads)
        qp = ops.gemm(t16[:, 0].contiguous(), P["pool_q"][0], bias=P["pool_q"][1]).view(B, 1, heads, dim // heads)
        o = ops.attention_small(qp, kvp[:, :, 0], kvp[:, :, 1], 1.0 / math.sqrt(dim // heads))
        pooled = ops.gemm(o.view(B, dim), P["pool_c"][0], bias=P["pool_c"][1], out_dtype=torch.float32)
        return prompt, pooled

    def _forward_precise(self, x):
        """forward() with fp32-grade activations: LayerNorms emit fp32, every GEMM reads its A operand as two 16-bit planes
        (ops.split16 → sx_gemm a_planes = 2, fp32 out), both attentions are fp32 (sx_attention_f32, causal = 0)."""
        P, dt, dim, heads, hd = self._pack(), self.dtype, self.dim, self.heads, self.dim_head
        B, n, _ = x.shape
        nq, inner, f32 = self.num_queries, self.heads * self.dim_head, torch.float32
        xf = x.to(self.device, f32)
        if self.normalize:
            from .image_ops import l2norm_dim1
            xf = l2norm_dim1(xf)
        lin = lambda a32, w, **kw: ops.gemm(ops.split16(a32, dt), w, a_planes=2, out_dtype=f32, **kw)
        xs = lin(xf.contiguous().view(B * n, -1), P["pin"][0], bias=P["pin"][1])                             # proj_in
        lat = P["latents"].unsqueeze(0).expand(B, nq, dim).contiguous().view(B * nq, dim)
        scale = 1.0 / math.sqrt(hd)
        for lw in P["layers"]:
            xn = ops.layernorm(xs, lw["n1"][0], lw["n1"][1], 1e-5, f32)
            ln = ops.layernorm(lat, lw["n2"][0], lw["n2"][1], 1e-5, f32)
            q = lin(ln, lw["wq"]).view(B, nq, heads, hd)
            kv = torch.empty((B, n + nq, 2 * inner), dtype=f32, device=self.device)                           # cat(x, latents)
            xn2, ln2 = ops.split16(xn, dt), ops.split16(ln, dt)
            for b in range(B):
                ops.gemm(xn2[b * n:(b + 1) * n], lw["wkv"], a_planes=2, out=kv[b, :n], out_dtype=f32)
                ops.gemm(ln2[b * nq:(b + 1) * nq], lw["wkv"], a_planes=2, out=kv[b, n:], out_dtype=f32)
            kv5 = kv.view(B, n + nq, 2, heads, hd)
            att = ops.attention_f32_full(q, kv5[:, :, 0], kv5[:, :, 1], scale, dt)                            # planes [B*nq, 2*inner]
            lat = ops.gemm(att, lw["wo"], a_planes=2, residual=lat, out_dtype=f32)
            h = lin(ops.layernorm(lat, lw["fn"][0], lw["fn"][1], 1e-5, f32), lw["w1"], act="gelu")
            lat = lin(h, lw["w3"], residual=lat)
        hid = ops.layernorm(lat, P["nout"][0], P["nout"][1], 1e-5, f32)                                       # norm_out
        prompt = lin(hid, P["proj"][0], bias=P["proj"][1]).view(B, nq, -1)
        mean = ops.avgpool_tokens(hid.view(B, nq, dim), nq)                                                   # AttentionPool2d
        t = torch.empty((B, nq + 1, dim), dtype=f32, device=self.device)
        t[:, 0:1] = mean
        t[:, 1:] = hid.view(B, nq, dim)
        t = ops.add(t, P["pool_pos"].unsqueeze(0).expand(B, nq + 1, dim).contiguous())
        hp = dim // heads
        kvp = lin(t.view(B * (nq + 1), dim), P["pool_kv"][0], bias=P["pool_kv"][1]).view(B, nq + 1, 2, heads, hp)
        qp = lin(t[:, 0].contiguous(), P["pool_q"][0], bias=P["pool_q"][1]).view(B, 1, heads, hp)
        o = ops.attention_f32_full(qp, kvp[:, :, 0], kvp[:, :, 1], 1.0 / math.sqrt(hp), dt)                   # planes [B, 2*dim]
        pooled = ops.gemm(o, P["pool_c"][0], a_planes=2, bias=P["pool_c"][1], out_dtype=f32)
        return prompt, pooled

    __call__ = forward


# ---------------------------------------------------------------------------------------------------------------
class EulerDiscreteScheduler:
    """SDXL scheduler config of diffusers' EulerDiscreteScheduler [ext]: scaled_linear betas 0.00085→0.012, 1000 train
    steps, `leading` spacing, steps_offset 1, epsilon prediction, linear sigma interpolation."""
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1, **_):
        self.config = dict(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                           steps_offset=steps_offset, timestep_spacing="leading", prediction_type="epsilon")
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.timesteps, self.sigmas = None, None

    @classmethod
    def from_pretrained(cls, path, subfolder=None):
        import json
        import os
        f = os.path.join(path, subfolder or "", "scheduler_config.json")
        j = json.load(open(f))
        assert j.get("timestep_spacing", "leading") == "leading" and j.get("beta_schedule") == "scaled_linear"
        return cls(j["num_train_timesteps"], j["beta_start"], j["beta_end"], j.get("steps_offset", 0))

    def set_timesteps(self, num_inference_steps, device=None):
        n = self.config["num_train_timesteps"]
        ac = self.alphas_cumprod.numpy()
        ratio = n // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.float32) + self.config["steps_offset"]
        sig = np.array(((1 - ac) / ac) ** 0.5)
        sig = np.interp(ts, np.arange(0, len(sig)), sig)
        sig = np.concatenate([sig, [0.0]]).astype(np.float32)
        self.timesteps, self.sigmas = torch.from_numpy(ts), torch.from_numpy(sig)
        self.num_inference_steps = num_inference_steps

    @property
    def init_noise_sigma(self):
        return float((self.sigmas.max() ** 2 + 1) ** 0.5)


# ---------------------------------------------------------------------------------------------------------------
class _DenoiseLoop:
    """Device-resident CFG + Euler loop around UNet.forward_nhwc, one HIP graph replay per step.

    ``comm`` with world == nb (2 for t2i, 3 for edit): CFG-parallel — rank r runs guidance branch r of every generation
    (UNet batch G instead of nb·G), the eps of all branches are all-gathered once per step (G·H·W·4 fp32 per rank) and
    the CFG + Euler update is replicated, so every rank carries identical latents (parallel.py)."""

    def __init__(self, unet, use_graph=True, comm=None):
        from .parallel import Comm
        self.unet, self.use_graph = unet, use_graph
        self.comm = comm or Comm()
        self._graph_key, self._graph, self._state = None, None, None
        self._ctx_static = None
        self.chains = 2                      # concurrent kernel chains per denoise step (1 = one serial chain)
        self.stagger = int(os.environ.get("SX_CHAIN_STAGGER", "0"))   # chain c starts when chain c-1 has passed this many progress marks
        self._side = None

    def run(self, mode, latents_nchw, prompt_embeds, pooled, time_ids, scheduler, num_steps, guidance_scale,
            image_guidance_scale=1.5, image_latents_nchw=None, trace=None):
        """mode 0 (t2i): prompt_embeds ordered [uncond, text]; mode 1 (edit): [text, image, uncond].
        latents_nchw: [G,4,H,W] already multiplied by init_noise_sigma, G >= 1 independent generations denoised as one
        UNet batch of nb·G samples ordered [branch][generation] (prompt_embeds / pooled / time_ids / image latents all
        follow that order). Returns fp32 [G,4,H,W]. ``trace``: optional dict {step index (1-based): None}; filled with a
        copy of the latents after that many steps (drift measurements; the pipeline's `callback` equivalent)."""
        unet = self.unet
        unet._pack()
        dev = unet.device
        nb = 2 if mode == 0 else 3
        G, Cl, H, W = latents_nchw.shape
        HW = H * W
        NB = nb * G
        assert prompt_embeds.shape[0] == NB and pooled.shape[0] == NB and time_ids.shape[0] == NB
        cin = unet.cfg["in_channels"]
        assert cin == (Cl if mode == 0 else 2 * Cl), f"UNet in_channels {cin} does not match mode {mode}"
        scheduler.set_timesteps(num_steps)
        ts_dev = scheduler.timesteps.to(dev)
        sig_dev = scheduler.sigmas.to(dev)
        key = (mode, G, H, W, num_steps, float(guidance_scale), float(image_guidance_scale), tuple(prompt_embeds.shape),
               tuple(pooled.shape), tuple(time_ids.shape), unet.dtype, str(dev), self.chains, self.stagger)
        if self._state is None or self._graph_key != key:
            self._state = dict(lat=torch.empty((G, HW, Cl), dtype=torch.float32, device=dev),
                               scaled=torch.zeros((NB, HW, cin), dtype=torch.float32, device=dev),
                               step=torch.zeros(1, dtype=torch.int32, device=dev),
                               ehs=torch.empty(prompt_embeds.shape, dtype=torch.float32, device=dev),
                               pooled=torch.empty(pooled.shape, dtype=torch.float32, device=dev),
                               tid=torch.empty(time_ids.shape, dtype=torch.float32, device=dev),
                               ts=torch.empty_like(ts_dev), sig=torch.empty_like(sig_dev))
            self._graph, self._graph_key = None, key
            self._ctx_static = None
        S = self._state
        comm = self.comm
        cfgp = comm.world > 1
        assert not cfgp or comm.world == nb, f"CFG-parallel needs one rank per guidance branch ({nb}), got {comm.world}"
        lo, hi = (comm.rank * G, (comm.rank + 1) * G) if cfgp else (0, NB)      # this rank's rows of the [branch][gen] batch
        S["ts"].copy_(ts_dev); S["sig"].copy_(sig_dev)
        S["ehs"].copy_(prompt_embeds.to(dev).float()); S["pooled"].copy_(pooled.to(dev).float())
        S["tid"].copy_(time_ids.to(dev).float())
        S["step"].zero_()
        ops.nchw_to_nhwc(latents_nchw.to(dev, torch.float32), dst=S["lat"])
        s0 = float(scheduler.sigmas[0])
        S["scaled"].view(nb, G, HW, cin)[:, :, :, :Cl] = S["lat"] / math.sqrt(s0 * s0 + 1.0)   # scale_model_input, step 0
        if mode == 1:
            il = image_latents_nchw.to(dev, torch.float32)                       # [3·G,4,H,W] = [enc, enc, 0] (:544-546)
            S["scaled"][:, :, Cl:] = il.permute(0, 2, 3, 1).reshape(NB, HW, Cl)
        unet._ctx_key = None
        ctx = unet.prepare_context(S["ehs"][lo:hi])                               # step-invariant cross-attn K/V
        if self._graph is not None and self._ctx_static is not None:
            for per_s, per_n in zip(self._ctx_static, ctx):                       # the graph holds these addresses
                for kv_s, kv_n in zip(per_s, per_n):
                    kv_s.copy_(kv_n)
            ctx = self._ctx_static
        else:
            self._ctx_static = ctx

        nloc = hi - lo
        chains = self.chains if (nloc % self.chains == 0 and nloc >= 2 * self.chains and unet.comm.world == 1) else 1
        if chains > 1 and (self._side is None or len(self._side) < chains - 1):
            self._side = [torch.cuda.Stream() for _ in range(chains - 1)]

        def step_body():
            temb = unet.time_embeddings(S["ts"], S["step"], S["pooled"][lo:hi], S["tid"][lo:hi], nloc)
            if chains == 1:
                eps = unet.forward_nhwc(S["scaled"][lo:hi], temb, ctx, nloc, H, W)
            else:
                # the samples of a UNet batch are independent: run them as `chains` concurrent kernel chains (forked HIP
                # streams, captured as parallel branches of the step's graph). A chain's memory-bound kernels and kernel
                # tails then fill under the other chain's MFMA-bound GEMMs instead of serialising behind them.
                per = nloc // chains
                eps = torch.empty((nloc, H * W, unet.cfg["out_channels"]), dtype=torch.float32, device=dev)
                main = torch.cuda.current_stream()
                fork = torch.cuda.Event()
                fork.record(main)
                parts = []
                # stagger: identical chains started together run the SAME kernel at the same time (two HBM-bound launches compete,
                # two MFMA-bound ones share the power budget); started `stagger` progress marks apart, a chain's HBM-bound launches
                # meet the other chain's MFMA-bound ones on the CU pool
                started = [torch.cuda.Event() for _ in range(chains - 1)] if self.stagger > 0 else None
                for c in range(chains):
                    sl = slice(lo + c * per, lo + (c + 1) * per)
                    cctx = [[kv[c * per:(c + 1) * per] for kv in per_t] for per_t in ctx]
                    st = main if c == 0 else self._side[c - 1]
                    if c:
                        st.wait_event(fork)
                        if started is not None:
                            st.wait_event(started[c - 1])
                    hook = None
                    if started is not None and c < chains - 1:
                        def hook(i, ev=started[c], s_=st):
                            if i == self.stagger:
                                ev.record(s_)
                    with torch.cuda.stream(st):
                        e = unet.forward_nhwc(S["scaled"][sl], temb[c * per:(c + 1) * per], cctx, per, H, W, on_mark=hook)
                        ops.copy2d(e.view(-1, e.shape[-1]), eps[c * per:(c + 1) * per].view(-1, e.shape[-1]), 0)
                        parts.append(e)
                for st in self._side[:chains - 1]:
                    main.wait_stream(st)
            if cfgp:
                eps = comm.all_gather(eps).reshape(NB, HW, -1)                    # [nb, G, HW, C] → [branch][generation]
            ops.cfg_euler_step(eps, S["lat"], S["scaled"], S["sig"], S["step"], nb, Cl, cin, guidance_scale,
                               image_guidance_scale, mode)
            ops.add_i32(S["step"], 1)

        def snap_trace(done):
            if trace is not None and done in trace:
                trace[done] = ops.nhwc_to_nchw(S["lat"], Cl, H, W).clone()

        if not self.use_graph or not comm.graph_safe or not getattr(unet, "comm", comm).graph_safe:
            for i in range(num_steps):
                step_body()
                snap_trace(i + 1)
        else:
            if self._graph is None:
                snap = {k: S[k].clone() for k in ("lat", "scaled", "step")}
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    step_body()                                                   # warm-up (allocator, lazy loads)
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    step_body()
                for k, v in snap.items():
                    S[k].copy_(v)
                self._graph = g
            for i in range(num_steps):
                self._graph.replay()
                snap_trace(i + 1)
        return ops.nhwc_to_nchw(S["lat"], Cl, H, W)


# ---------------------------------------------------------------------------------------------------------------
def vae_image_preprocess(image, vae_scale_factor=8):
    """``VaeImageProcessor.preprocess`` [ext diffusers 0.25.0] as the edit pipeline calls it (pipeline…:823, no explicit
    height/width): PIL / ndarray / tensor (or lists of them) → float tensor [B,C,H,W]; PIL images are resized (lanczos) down
    to a multiple of ``vae_scale_factor``; RGB in [0,1] is mapped to [-1,1] (input that is already negative is left alone);
    a 4-channel tensor is taken to be latents and returned untouched."""
    from PIL import Image
    if isinstance(image, (Image.Image, np.ndarray)) or (torch.is_tensor(image) and image.ndim == 3):
        image = [image]
    if isinstance(image, (list, tuple)):
        if isinstance(image[0], Image.Image):
            arrs = []
            for im in image:
                w, h = im.size
                w, h = w - w % vae_scale_factor, h - h % vae_scale_factor
                if (w, h) != im.size:
                    im = im.resize((w, h), resample=Image.LANCZOS)
                arrs.append(np.asarray(im.convert("RGB"), dtype=np.float32) / 255.0)
            image = torch.from_numpy(np.stack(arrs, 0)).permute(0, 3, 1, 2)
        elif isinstance(image[0], np.ndarray):
            arrs = [a if a.ndim == 4 else a[None] for a in image]
            image = torch.from_numpy(np.concatenate(arrs, 0)).permute(0, 3, 1, 2).float()
        else:
            image = torch.cat([t if t.ndim == 4 else t[None] for t in image], dim=0)
    if not torch.is_tensor(image) or image.ndim != 4:
        raise ValueError(f"`image` has to be a PIL image, ndarray, tensor or a list of them but is {type(image)}")
    if image.shape[1] == 4:
        return image
    image = image.float()
    return 2.0 * image - 1.0 if float(image.min()) >= 0 else image


class SDXLAdapter:
    def __init__(self, unet, resampler, full_ft=False, vit_down=False):
        self.unet, self.resampler = unet, resampler
        self.full_ft, self.vit_down = full_ft, vit_down
        self.device, self.dtype = None, torch.float16
        self.visual_encoder = self.image_transform = self.discrete_model = self.vae = self.scheduler = None
        self._neg_cache = {}
        self._loop = None
        self.use_graph = True

    @classmethod
    def from_pretrained(cls, unet, resampler, pretrained_model_path=None, **kwargs):
        model = cls(unet=unet, resampler=resampler, **kwargs)
        if pretrained_model_path is not None:
            ckpt = torch.load(pretrained_model_path, map_location="cpu")          # adapter_modules.py:62-65
            model.load_state_dict(ckpt)
        return model

    def load_state_dict(self, sd, strict=True):
        """Checkpoint keys: resampler.* and unet.* (second stage: 8-channel unet.conv_in). The unet.* entries OVERLAY the
        weights the UNet already holds, like the reference's `load_state_dict(ckpt, strict=False)` over the SDXL-base
        UNet (adapter_modules.py:62-65): a first-stage checkpoint may carry only the trained to_k / to_v tensors."""
        self.resampler.load_state_dict(sd, prefix="resampler.", strict=strict)
        if any(k.startswith("unet.") for k in sd):
            self.unet.load_state_dict(sd, prefix="unet.", strict=strict, merge=True)
        return [], []

    def to(self, device=None, dtype=None):
        if device is not None:
            self.device = torch.device(device)
        if dtype is not None:
            self.dtype = dtype
        self.unet.to(self.device, self.dtype)
        self.resampler.to(self.device, self.dtype)
        self._neg_cache = {}
        if self._loop is not None:
            self._loop._graph = self._loop._state = self._loop._ctx_static = None
        return self

    def eval(self):
        return self

    @torch.no_grad()
    def forward(self, noisy_latents, timesteps, image_embeds, text_embeds, noise, time_ids):
        """adapter_modules.py:39-52 (the training-time forward, inference-only here: no autograd): resampler → ONE UNet
        forward conditioned on the resampled image tokens + pooled vector (`text_embeds` is accepted and ignored, exactly
        like the reference) → {'total_loss': mse(noise_pred, noise), 'noise_pred': [B, 4, h, w]}."""
        image_embeds, pooled_image_embeds = self.resampler(image_embeds)
        unet_added_conditions = {"time_ids": time_ids, "text_embeds": pooled_image_embeds}
        noise_pred = self.unet(noisy_latents, timesteps, image_embeds, added_cond_kwargs=unet_added_conditions).sample
        loss = torch.nn.functional.mse_loss(noise_pred.float(), noise.to(noise_pred.device).float(), reduction="mean")
        return {"total_loss": loss, "noise_pred": noise_pred}

    __call__ = forward

    def encode_image_embeds(self, image_embeds):
        return self.resampler(image_embeds)                                         # adapter_modules.py:54-57

    def init_pipe(self, vae, scheduler, visual_encoder, image_transform, discrete_model=None, dtype=torch.float16,
                  device='cuda'):
        self.device, self.dtype = torch.device(device), dtype
        self.vae, self.scheduler = vae, scheduler
        if vae is not None and getattr(vae, "device", self.device) is None and hasattr(vae, "to"):
            vae.to(self.device, self.dtype)          # the edit pipeline moves its modules itself (adapter_modules.py:243)
        self.visual_encoder = visual_encoder.to(self.device, dtype=self.dtype)
        self.discrete_model = discrete_model
        self.image_transform = image_transform
        self.to(self.device, self.dtype)
        self._neg_cache = {}                                                        # new encoder / dtype → new negatives
        self._loop = _DenoiseLoop(self.unet, self.use_graph, comm=getattr(self, "comm", None))

    def _negative_embeds(self, image_size, pooled):
        """ViT features of an all-zero image: constant per model → cached (the reference recomputes a full ViT forward
        on every call, adapter_modules.py:109-116)."""
        key = (image_size, pooled)
        if key not in self._neg_cache:
            z = torch.zeros(1, 3, image_size, image_size, device=self.device)
            e = self.visual_encoder(z).float()
            if pooled:
                e = ops.avgpool_tokens(e.contiguous(), 4)                           # avg_pool1d(k=4,s=4), :112-115
            self._neg_cache[key] = e
        return self._neg_cache[key]

    @torch.no_grad()
    def get_image_embeds(self, image_pil=None, image_tensor=None, image_embeds=None, return_negative=True, image_size=448):
        assert int(image_pil is not None) + int(image_tensor is not None) + int(image_embeds is not None) == 1
        if image_pil is not None:
            image_tensor = self.image_transform(image_pil).unsqueeze(0)
        if image_tensor is not None:
            image_tensor = image_tensor.to(self.device, torch.float32)
            if return_negative:
                image_tensor = torch.cat([image_tensor, torch.zeros_like(image_tensor)], dim=0)   # :103-106
            image_embeds = self.visual_encoder(image_tensor).float()                # 256 tokens, NO pooling (:108)
        elif return_negative:
            neg = self._negative_embeds(image_size, self.vit_down)
            ie = image_embeds.to(self.device).float()
            image_embeds = torch.cat([ie, neg.expand(ie.shape[0], -1, -1)], dim=0)        # :116 (one negative per sample)
        if self.discrete_model is not None:
            image_embeds = self.discrete_model.encode_image_embeds(image_embeds)    # identity (discrete_models.py:16-17)
        prompt, pooled = self.encode_image_embeds(image_embeds)
        if return_negative:
            prompt, prompt_neg = prompt.chunk(2)
            pooled, pooled_neg = pooled.chunk(2)
        else:
            prompt_neg = pooled_neg = None
        return prompt, prompt_neg, pooled, pooled_neg

    def _time_ids(self, height, width, n):
        return torch.tensor([[height, width, 0, 0, height, width]] * n, dtype=torch.float32)   # _get_add_time_ids :554-566

    def _noise(self, seed, height, width, n=1):
        """One [1,4,h,w] noise tensor per generation; `seed` may be an int (generation i uses seed + i) or a list."""
        out = []
        for i in range(n):
            sd = None if seed is None else (seed[i] if isinstance(seed, (list, tuple)) else seed + i)
            g = torch.Generator(self.device).manual_seed(sd) if sd is not None else None
            out.append(torch.randn((1, 4, height // 8, width // 8), generator=g, device=self.device, dtype=torch.float32))
        return torch.cat(out, dim=0)

    def _finish(self, latents, output_type):
        """pipeline…:965-986: "latent" → latents; otherwise VAE-decode (latents / scaling_factor) and post-process like
        VaeImageProcessor.postprocess [ext]: "pt" → [B,3,H,W] in [0,1]; "np" → [B,H,W,3]; "pil" → list of PIL images."""
        if output_type is None:                                                      # the reference's default is "pil"
            output_type = "pil" if self.vae is not None else "latent"
        if output_type == "latent":
            return latents
        if self.vae is None:
            raise RuntimeError(f"output_type={output_type!r} needs a VAE (init_pipe(vae=...)); use output_type='latent'")
        scaling = getattr(getattr(self.vae, "config", None), "scaling_factor", 0.13025)
        img = self.vae.decode(latents / scaling, return_dict=False)[0]
        if output_type == "raw":                                                     # decoder output, no post-processing
            return img
        if output_type in ("pil", "u8"):                                             # fused denormalise → uint8 HWC kernel
            from .image_ops import images_to_pil, images_to_u8
            return images_to_pil(img.float()) if output_type == "pil" else images_to_u8(img.float())   # u8: stays in HBM
        img = (img / 2 + 0.5).clamp(0, 1)
        if output_type == "pt":
            return img
        if output_type == "np":
            return img.permute(0, 2, 3, 1).float().cpu().numpy()
        raise ValueError(f"unknown output_type {output_type!r}")

    def generate(self, image_pil=None, image_tensor=None, image_embeds=None, seed=None, height=1024, width=1024,
                 guidance_scale=7.5, num_inference_steps=30, input_image_size=448, output_type=None, latents=None,
                 **kwargs):
        if image_pil is not None:
            from PIL import Image
            assert isinstance(image_pil, Image.Image)                                # adapter_modules.py:143-144
        pe, pe_neg, pool, pool_neg = self.get_image_embeds(image_pil=image_pil, image_tensor=image_tensor,
                                                           image_embeds=image_embeds, return_negative=True,
                                                           image_size=input_image_size)
        self.scheduler.set_timesteps(num_inference_steps)
        G = pe.shape[0]                                                              # generations in this call
        if latents is None:
            latents = self._noise(seed, height, width, G)
        latents = latents.to(self.device, torch.float32) * self.scheduler.init_noise_sigma
        ehs = torch.cat([pe_neg, pe], dim=0)                                         # order [uncond, text] x G
        pooled = torch.cat([pool_neg, pool], dim=0)
        out = self._loop.run(0, latents, ehs, pooled, self._time_ids(height, width, 2 * G), self.scheduler,
                             num_inference_steps, guidance_scale)
        return self._finish(out, output_type)


class SDXLAdapterWithLatentImage(SDXLAdapter):
    """Edit variant (adapter_modules.py:172-287): 8-channel UNet input, 3-way guidance."""

    def __init__(self, unet, resampler, full_ft=False, set_trainable_late=False, vit_down=False):
        super().__init__(unet, resampler, full_ft=full_ft, vit_down=vit_down)
        if not set_trainable_late:
            self.set_trainable()

    def set_trainable(self):
        """The inference-relevant part of adapter_modules.py:186-198: conv_in grows to 8 input channels, the new ones
        zero-initialised, the first four copied — so a checkpoint without unet.conv_in still yields a valid edit UNet."""
        self.unet.expand_conv_in(8)

    @classmethod
    def from_pretrained(cls, unet, resampler, pretrained_model_path=None, set_trainable_late=False, **kwargs):
        model = cls(unet=unet, resampler=resampler, set_trainable_late=set_trainable_late, **kwargs)
        if pretrained_model_path is not None:
            ckpt = torch.load(pretrained_model_path, map_location="cpu")          # adapter_modules.py:215-218
            model.load_state_dict(ckpt)
        if set_trainable_late:
            model.set_trainable()
        return model

    def init_pipe(self, vae, scheduler, visual_encoder, image_transform, dtype=torch.float16, device='cuda', **kw):
        return super().init_pipe(vae, scheduler, visual_encoder, image_transform, discrete_model=None, dtype=dtype,
                                 device=device)

    def generate(self, image_pil=None, image_tensor=None, image_embeds=None, latent_image=None, seed=42, height=1024,
                 width=1024, guidance_scale=7.5, num_inference_steps=30, input_image_size=448,
                 image_guidance_scale=1.5, output_type=None, latents=None, image_latents=None, **kwargs):
        if image_pil is not None:
            from PIL import Image
            assert isinstance(image_pil, Image.Image)                                # adapter_modules.py:261-262
        pe, pe_neg, pool, pool_neg = self.get_image_embeds(image_pil=image_pil, image_tensor=image_tensor,
                                                           image_embeds=image_embeds, return_negative=True,
                                                           image_size=input_image_size)
        self.scheduler.set_timesteps(num_inference_steps)
        G = pe.shape[0]
        if latents is None:
            latents = self._noise(seed, height, width, G)
        latents = latents.to(self.device, torch.float32) * self.scheduler.init_noise_sigma
        if image_latents is None:
            if latent_image is None:
                image_latents = torch.zeros(G, 4, height // 8, width // 8)          # pipeline…:909-910 (no source image)
            else:
                src = vae_image_preprocess(latent_image)                            # pipeline…:823
                if src.shape[1] == 4:                                               # already latents (:500-505)
                    image_latents = src
                else:
                    if self.vae is None:
                        raise RuntimeError("an RGB `latent_image` needs a VAE with encoder weights (init_pipe(vae=...)); "
                                           "or pass 4-channel latents = vae.encode(img).latent_dist.mode() (NOT scaled, :523)")
                    image_latents = self.vae.encode(src).latent_dist.mode()
                if image_latents.shape[0] != G:                                     # :529-541 duplicate per prompt
                    if G % image_latents.shape[0] != 0:
                        raise ValueError(f"Cannot duplicate `image` of batch size {image_latents.shape[0]} to {G} prompts.")
                    image_latents = torch.cat([image_latents] * (G // image_latents.shape[0]), dim=0)
        il = image_latents.to(self.device, torch.float32)
        il3 = torch.cat([il, il, torch.zeros_like(il)], dim=0)                       # [img, img, 0]  (:544-546)
        ehs = torch.cat([pe, pe_neg, pe_neg], dim=0)                                 # order [text, image, uncond] (:884)
        pooled = torch.cat([pool, pool_neg, pool_neg], dim=0)
        out = self._loop.run(1, latents, ehs, pooled, self._time_ids(height, width, 3 * G), self.scheduler,
                             num_inference_steps, guidance_scale, image_guidance_scale, il3)
        return self._finish(out, output_type)
