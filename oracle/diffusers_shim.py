"""Stand-in ``diffusers`` package so that the REFERENCE's own de-tokenizer code runs on CPU — TEST INFRASTRUCTURE.

``diffusers==0.25.0`` (reference requirements.txt:4) is not installed and not under /root/reference, but two reference
files that sit ON the hot path only import it for base classes, type names and small helpers:

  * src/models/detokenizer/pipeline_stable_diffusion_xl_t2i_edit.py  (the edit pipeline; ``__call__`` :618-994 with the
    denoise loop :900-963 is reference code)
  * src/models/detokenizer/adapter_modules.py                         (``SDXLAdapter*``: get_image_embeds :96-130,
    generate :132-169 / :249-287, set_trainable :186-209)

``install()`` registers minimal modules under ``sys.modules['diffusers…']`` so both import unchanged, and this file
provides duck-typed components the reference code can drive:

  * ``OracleUNet``            nn.Module whose parameters carry diffusers' UNet2DConditionModel key names and whose
                              forward is oracle/restated_unet.unet_forward (restated, third-party → still UNPINNED)
  * ``EulerDiscreteScheduler`` restatement of diffusers' scheduler API used by the pipelines (set_timesteps,
                              scale_model_input, step, init_noise_sigma) — third-party, unpinned
  * ``OracleVAE``             AutoencoderKL duck type over oracle/restated_vae (third-party, unpinned)
  * ``VaeImageProcessor``     preprocess / postprocess [ext] as used at pipeline…:823,986
  * ``StableDiffusionXLPipeline`` the t2i ``__call__`` [ext] restated for the arguments adapter_modules.py:156-167
                              passes (prompt_embeds given, text encoders None)

What this buys: the reference's OWN edit loop, image-latent preparation, CFG ordering, sigma-space hack, adapter front
end (three get_image_embeds branches, pooling asymmetry, 8-channel conv_in surgery) are EXECUTED, not restated, when the
goldens ``tests/golden/{edit,t2i}_mini.npz`` are produced (oracle/gen_golden.py). Only the network bodies called from
that code (UNet / VAE / scheduler arithmetic) remain restatements of the third-party package.
"""
import importlib.machinery
import sys
import types

import numpy as np
import torch
import torch.nn as nn

from . import restated_unet as ru
from . import restated_vae as rv


# ---------------------------------------------------------------------------------------------------------------
# duck-typed components
# ---------------------------------------------------------------------------------------------------------------
class _Node(nn.Module):
    """Parameter container: 'a.b.weight' becomes self.a.b.weight so state_dict() round-trips diffusers' key names."""


def _tree_from_sd(root, sd):
    for name, t in sd.items():
        parts = name.split(".")
        node = root
        for p in parts[:-1]:
            if p not in node._modules:
                node.add_module(p, _Node())
            node = node._modules[p]
        node.register_parameter(parts[-1], nn.Parameter(t.detach().clone().float(), requires_grad=False))


class OracleUNet(nn.Module):
    """UNet2DConditionModel duck type (attributes read by the reference: config.{sample_size, addition_time_embed_dim,
    in_channels}, add_embedding.linear_1.in_features, conv_in (nn.Conv2d, replaced by set_trainable), dtype,
    register_to_config; call forms pipeline…:915-922 and adapter_modules.py:45)."""

    def __init__(self, cfg, sd, sample_size=16):
        super().__init__()
        self.cfg = dict(cfg)
        rest = {k: v for k, v in sd.items() if not k.startswith("conv_in.")}
        _tree_from_sd(self, rest)
        w = sd["conv_in.weight"]
        self.conv_in = nn.Conv2d(w.shape[1], w.shape[0], 3, 1, 1)
        with torch.no_grad():
            self.conv_in.weight.copy_(w)
            self.conv_in.bias.copy_(sd["conv_in.bias"])
        self.config = types.SimpleNamespace(sample_size=sample_size, in_channels=cfg["in_channels"],
                                            addition_time_embed_dim=cfg["addition_time_embed_dim"])
        self.add_embedding.linear_1.in_features = self.add_embedding.linear_1.weight.shape[1]

    def register_to_config(self, **kw):
        for k, v in kw.items():
            setattr(self.config, k, v)

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    def forward(self, sample, timestep, encoder_hidden_states, cross_attention_kwargs=None, added_cond_kwargs=None,
                return_dict=True, **_):
        sd = {k: v for k, v in self.state_dict().items()}
        cfg = dict(self.cfg, in_channels=self.conv_in.in_channels)
        out = ru.unet_forward(sd, cfg, sample, timestep, encoder_hidden_states, added_cond_kwargs["text_embeds"],
                              added_cond_kwargs["time_ids"])
        return (out,) if not return_dict else types.SimpleNamespace(sample=out)


class EulerDiscreteScheduler:
    """diffusers 0.25.0 EulerDiscreteScheduler [ext] with the SDXL scheduler_config.json values (scaled_linear betas,
    `leading` spacing, steps_offset 1, epsilon prediction, linear interpolation, no Karras sigmas, gamma = 0)."""
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1):
        self.config = types.SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start,
                                            beta_end=beta_end, steps_offset=steps_offset)
        self.timesteps = self.sigmas = None
        self._step_index = None

    def set_timesteps(self, num_inference_steps, device=None):
        c = self.config
        ts, sig, _ = ru.euler_tables(num_inference_steps, c.num_train_timesteps, c.beta_start, c.beta_end, c.steps_offset)
        self.timesteps, self.sigmas = ts, sig
        self.num_inference_steps = num_inference_steps
        self._step_index = None

    @property
    def init_noise_sigma(self):
        return (self.sigmas.max() ** 2 + 1) ** 0.5            # `leading` spacing branch

    def _index(self, t):
        return int((self.timesteps == t).nonzero()[0].item())

    def scale_model_input(self, sample, timestep):
        if self._step_index is None:
            self._step_index = self._index(timestep)
        sigma = self.sigmas[self._step_index]
        return sample / ((sigma ** 2 + 1) ** 0.5)

    def step(self, model_output, timestep, sample, return_dict=True):
        if self._step_index is None:
            self._step_index = self._index(timestep)
        sigma = self.sigmas[self._step_index]                 # gamma = 0 → sigma_hat = sigma
        pred_original_sample = sample - sigma * model_output  # epsilon prediction
        derivative = (sample - pred_original_sample) / sigma
        dt = self.sigmas[self._step_index + 1] - sigma
        prev_sample = sample + derivative * dt
        self._step_index += 1
        return (prev_sample,) if not return_dict else types.SimpleNamespace(prev_sample=prev_sample)


class _LatentDist:
    def __init__(self, mean):
        self._mean = mean

    def mode(self):
        return self._mean


class OracleVAE(nn.Module):
    """AutoencoderKL duck type over oracle/restated_vae (decode; encode().latent_dist.mode())."""

    def __init__(self, cfg, sd_dec, sd_enc=None, scaling_factor=0.13025, force_upcast=True):
        super().__init__()
        self.cfg, self.sd_dec, self.sd_enc = cfg, sd_dec, sd_enc
        self.config = types.SimpleNamespace(block_out_channels=tuple(cfg["block_out_channels"]),
                                            latent_channels=cfg.get("latent_channels", 4),
                                            scaling_factor=scaling_factor, force_upcast=force_upcast)
        self.post_quant_conv = nn.Conv2d(1, 1, 1)            # only .parameters() dtype is read (pipeline…:969)

    @property
    def dtype(self):
        return torch.float32

    def decode(self, z, return_dict=True):
        img = rv.vae_decode(self.sd_dec, self.cfg, z.float())
        return (img,) if not return_dict else types.SimpleNamespace(sample=img)

    def encode(self, x):
        return types.SimpleNamespace(latent_dist=_LatentDist(rv.vae_encode_mode(self.sd_enc, self.cfg, x.float())))


class VaeImageProcessor:
    """diffusers.image_processor.VaeImageProcessor [ext] defaults: do_resize to a multiple of vae_scale_factor
    (lanczos), do_normalize to [-1, 1]; postprocess = denormalise → clamp → NHWC → uint8 PIL."""

    def __init__(self, vae_scale_factor=8, do_resize=True, do_normalize=True, resample="lanczos"):
        self.f, self.do_resize, self.do_normalize = vae_scale_factor, do_resize, do_normalize

    def preprocess(self, image, height=None, width=None):
        import PIL.Image
        if isinstance(image, PIL.Image.Image):
            image = [image]
        if isinstance(image, list) and isinstance(image[0], PIL.Image.Image):
            out = []
            for im in image:
                w, h = im.size
                w, h = (width or w), (height or h)
                w, h = w - w % self.f, h - h % self.f
                if self.do_resize:
                    im = im.resize((w, h), resample=PIL.Image.LANCZOS)
                out.append(np.array(im.convert("RGB")).astype(np.float32) / 255.0)
            t = torch.from_numpy(np.stack(out, 0)).permute(0, 3, 1, 2)
        elif isinstance(image, np.ndarray):
            t = torch.from_numpy(image if image.ndim == 4 else image[None]).permute(0, 3, 1, 2).float()
        else:
            t = image if image.ndim == 4 else image[None]
            if t.shape[1] == 4:                                # already latents: passed through untouched
                return t
        if self.do_normalize and t.min() >= 0:
            t = 2.0 * t - 1.0
        return t

    def postprocess(self, image, output_type="pil", do_denormalize=None):
        if output_type == "latent":
            return image
        image = (image / 2 + 0.5).clamp(0, 1)
        if output_type == "pt":
            return image
        arr = image.cpu().permute(0, 2, 3, 1).float().numpy()
        if output_type == "np":
            return arr
        import PIL.Image
        return [PIL.Image.fromarray((a * 255).round().astype("uint8")) for a in arr]


class _Output:
    def __init__(self, images):
        self.images = images


class DiffusionPipeline:
    """The few DiffusionPipeline services the reference pipeline uses (register_modules / register_to_config /
    _execution_device / progress_bar / maybe_free_model_hooks / to)."""

    def __init__(self):
        self.config = types.SimpleNamespace()

    def register_modules(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    def register_to_config(self, **kw):
        for k, v in kw.items():
            setattr(self.config, k, v)

    @property
    def _execution_device(self):
        return torch.device("cpu")

    def to(self, *a, **k):
        return self

    def maybe_free_model_hooks(self):
        pass

    def progress_bar(self, iterable=None, total=None):
        class _PB:
            def __enter__(s):
                return s

            def __exit__(s, *e):
                return False

            def update(s, *a):
                pass
        return _PB()


class StableDiffusionXLPipeline(DiffusionPipeline):
    """StableDiffusionXLPipeline.__call__ [ext diffusers 0.25.0] for the call adapter_modules.py:156-167 makes:
    prompt / negative embeddings given, text encoders None, CFG order [uncond, text], Euler steps."""

    def __init__(self, vae, text_encoder, text_encoder_2, tokenizer, tokenizer_2, unet, scheduler, **_):
        super().__init__()
        self.register_modules(vae=vae, unet=unet, scheduler=scheduler, text_encoder=None, text_encoder_2=None)
        self.vae_scale_factor = 2 ** (len(vae.config.block_out_channels) - 1) if vae is not None else 8
        self.image_processor = VaeImageProcessor(self.vae_scale_factor)

    @torch.no_grad()
    def __call__(self, prompt_embeds, negative_prompt_embeds, pooled_prompt_embeds, negative_pooled_prompt_embeds,
                 guidance_scale=5.0, num_inference_steps=50, generator=None, height=None, width=None, latents=None,
                 output_type="pil", callback=None, callback_steps=1, **_):
        B = prompt_embeds.shape[0]
        height = height or self.unet.config.sample_size * self.vae_scale_factor
        width = width or self.unet.config.sample_size * self.vae_scale_factor
        self.scheduler.set_timesteps(num_inference_steps)
        shape = (B, self.unet.config.in_channels, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if latents is None:
            latents = torch.randn(shape, generator=generator, dtype=prompt_embeds.dtype)
        latents = latents * self.scheduler.init_noise_sigma
        add_time_ids = torch.tensor([[height, width, 0, 0, height, width]], dtype=prompt_embeds.dtype).repeat(B, 1)
        ehs = torch.cat([negative_prompt_embeds, prompt_embeds], dim=0)
        te = torch.cat([negative_pooled_prompt_embeds, pooled_prompt_embeds], dim=0)
        tid = torch.cat([add_time_ids, add_time_ids], dim=0)
        for i, t in enumerate(self.scheduler.timesteps):
            inp = self.scheduler.scale_model_input(torch.cat([latents] * 2), t)
            eps = self.unet(inp, t, encoder_hidden_states=ehs, added_cond_kwargs={"text_embeds": te, "time_ids": tid},
                            return_dict=False)[0]
            eu, et = eps.chunk(2)
            eps = eu + guidance_scale * (et - eu)
            latents = self.scheduler.step(eps, t, latents, return_dict=False)[0]
            if callback is not None and i % callback_steps == 0:
                callback(i, t, latents)
        if output_type == "latent":
            return _Output(latents)
        image = self.vae.decode(latents / self.vae.config.scaling_factor, return_dict=False)[0]
        return _Output(self.image_processor.postprocess(image, output_type=output_type))


# ---------------------------------------------------------------------------------------------------------------
# sys.modules registration
# ---------------------------------------------------------------------------------------------------------------
def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install():
    if "diffusers" in sys.modules and getattr(sys.modules["diffusers"], "_seedx_shim", False):
        return

    class _Empty:
        pass

    def _named(n):
        return type(n, (), {})

    logging = types.SimpleNamespace(get_logger=lambda name=None: __import__("logging").getLogger(name or "diffusers"))
    d = _mod("diffusers", _seedx_shim=True, StableDiffusionXLPipeline=StableDiffusionXLPipeline,
             EulerDiscreteScheduler=EulerDiscreteScheduler, AutoencoderKL=OracleVAE, UNet2DConditionModel=OracleUNet)
    d.image_processor = _mod("diffusers.image_processor", PipelineImageInput=object, VaeImageProcessor=VaeImageProcessor)
    d.loaders = _mod("diffusers.loaders", FromSingleFileMixin=_named("FromSingleFileMixin"),
                     StableDiffusionXLLoraLoaderMixin=_named("StableDiffusionXLLoraLoaderMixin"),
                     TextualInversionLoaderMixin=_named("TextualInversionLoaderMixin"))
    d.models = _mod("diffusers.models", AutoencoderKL=OracleVAE, UNet2DConditionModel=OracleUNet)
    _mod("diffusers.models.attention_processor", AttnProcessor2_0=_named("AttnProcessor2_0"),
         LoRAAttnProcessor2_0=_named("LoRAAttnProcessor2_0"), LoRAXFormersAttnProcessor=_named("LoRAXFormersAttnProcessor"),
         XFormersAttnProcessor=_named("XFormersAttnProcessor"))
    _mod("diffusers.models.lora", adjust_lora_scale_text_encoder=lambda *a, **k: None)
    _mod("diffusers.schedulers", KarrasDiffusionSchedulers=object)
    _mod("diffusers.utils", USE_PEFT_BACKEND=False, deprecate=lambda *a, **k: None,
         is_invisible_watermark_available=lambda: False, is_torch_xla_available=lambda: False, logging=logging,
         replace_example_docstring=lambda doc: (lambda fn: fn), scale_lora_layers=lambda *a, **k: None)

    def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
        return torch.randn(shape, generator=generator, dtype=dtype)

    _mod("diffusers.utils.torch_utils", randn_tensor=randn_tensor)
    _mod("diffusers.pipelines")
    _mod("diffusers.pipelines.pipeline_utils", DiffusionPipeline=DiffusionPipeline)
    _mod("diffusers.pipelines.stable_diffusion_xl")
    _mod("diffusers.pipelines.stable_diffusion_xl.pipeline_output", StableDiffusionXLPipelineOutput=_Output)


def reference_adapters():
    """The reference's adapter classes and edit pipeline, imported from /root/reference over the stand-in package."""
    from . import refshim
    refshim.install()
    install()
    from src.models.detokenizer import adapter_modules, pipeline_stable_diffusion_xl_t2i_edit as pipe
    return adapter_modules.SDXLAdapter, adapter_modules.SDXLAdapterWithLatentImage, \
        pipe.StableDiffusionXLText2ImageAndEditPipeline
