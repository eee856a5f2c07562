#!/usr/bin/env python
"""bench.py — end-to-end SEED-X generations/sec on MI355X (BASELINE.json metric).

Default workload (`--config 0`, the headline metric: image in → text + 1024 px image out, BASELINE configs 2 + 3
composed). One "step" = `--batch` independent generations processed together on one GPU, every input already resident
in HBM as the RAW uint8 image and the prompt token ids:
    448x448x3 uint8 image → any-res tiling + resize + CLIP normalise ON THE GPU (2 crops) → ViT-G/448 → input resampler
    → Llama-13B-dim prefill (165 tokens, all requests as one M = batch·165 pass) → greedy decode of 128 new tokens
    (61 text tokens with lm_head + logits rule, then <img> + 64 forced image tokens + </img>; EOS disabled so the
    length is fixed) → output resampler → ResamplerXLV2 (CFG batch 2; the all-zero-image negative ViT features are a
    per-model constant and cached) → 50-step SDXL UNet CFG(7.5)+Euler at 128x128 latents (= 1024x1024 px) → SDXL VAE
    decoder → uint8 [1024, 1024, 3] image (`--no-vae` stops at the latents).
`--config 1..5` run BASELINE.json's five configs (see CONFIGS below); they are parity / coverage workloads, the driver's
bench line is config 0.

Synthetic data: seeded random image / prompt ids, random-init weights of the real architecture (no checkpoints exist
here). N > 1 GPUs: one process per GPU, independent generations per rank (no data-path collective; weak scaling);
value = total generations of all ranks / max-over-ranks wall time. `python bench.py --gpus N` starts the N ranks itself
(re-exec under torch.distributed.run); under an external launcher (WORLD_SIZE set) it joins that job.

Usage: python bench.py --gpus N --steps K --warmup W [--config C]   (prints ONE JSON line on rank 0)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_TFLOPS_16BIT = 2500.0   # dense bf16/fp16 MFMA peak, MI355X_MICROARCH.md
# algorithmic FLOPs
FLOP_VIT_CROP = 4.219e12
FLOP_LLM_TOKEN = 25.71e9
FLOP_UNET_SAMPLE = 6.747e12
FLOP_VAE_DECODE = 10.47e12   # SDXL VAE decoder at 128x128 latents (conv/linear/attention MACs x 2)

# rel-L2 bounds vs the fp32 oracle ASSERTED by the full-size GPU parity tests, per module. "measured" = the figure of the
# last full run of the named test (profiles/r6_fulldepth.log, r6_pytest_gpu.log; the bound that counts is "asserted": the tests fail
# above it whatever this table says); the oracle sees the same inputs and the weights a 16-bit checkpoint holds.
PARITY_BOUND = {
    "fp16": {
        "metric": "rel-L2 vs the fp32 oracle (oracle/restated*.py) on the same inputs; every row is an assert in the named GPU test",
        "ViT-G/448 features, all 48 layers, B = 2 crops": {"asserted": 1e-3, "measured": 3.2e-4, "measured_unrounded_fp32_weights": 8.1e-4,
                                                             "test": "tests/test_fulldepth_gpu.py::test_vit_g_48_layers"},
        "LLM logits (all positions) and final-norm states, all 40 layers at 13B dims: 165-token prefill, decode steps 1 / 64 / 128 "
        "(precise mode = the default for every lock-step batch size up to 32; fp16: mixed KV cache, k fp32 / v 16-bit)": {
            "asserted": 1e-3, "measured": 5.6e-4, "measured_all_fp32_cache": 2.9e-5,
            "test": "tests/test_fulldepth_gpu.py::test_llama_13b_40_layers_prefill_and_128_decode_steps, "
                    "::test_llama_13b_40_layers_all_fp32_cache_floor (asserted 1e-4)"},
        "SDXL UNet latents, 2.57 B parameters at 128x128: one forward 4-ch CFG-2 / 8-ch Bc = 3; 50-step CFG-7.5 loop at steps 1 / 10 / 25 / 50": {
            "asserted": 1e-3, "measured": 8.6e-4, "test": "tests/test_fullsize_gpu.py::test_unet_full_sdxl_forward, "
            "tests/test_fullsize2_gpu.py::test_unet_full_8ch_bc3_forward, ::test_full_size_50_step_t2i_loop_drift"},
        "config-0 generation at full size and depth, every stage against the oracle stage on the SAME inputs (ViT | resamplers + LLM | "
        "ResamplerXLV2 + 50 UNet steps | VAE), every module on the weights a 16-bit checkpoint holds": {
            "asserted": 1e-3, "measured": [3.1e-4, 3.4e-4, 5.5e-4, 6.9e-6], "measured_unrounded_fp32_weights": [8.3e-4, 6.2e-4, 9.0e-4, 4.3e-4],
                                                  "test": "tests/test_fulldepth_gpu.py::test_config0_one_generation_end_to_end"},
        "same generation, oracle chain on its OWN intermediates (the stages' errors compound)": {"asserted": 2.5e-3, "measured": 5.6e-4,
                                                                                               "test": "same"},
        "SDXL VAE decode / encode at 1024 px (fp32-grade mode: two fp16 activation planes x the checkpoint's exact fp16 weights)": {"asserted": 1e-4, "measured": 1.4e-5,
                                                                  "test": "tests/test_fullsize2_gpu.py::test_vae_full_config_1024px"},
        "the LLM's plain 16-bit flow (precise=False / SX_LLM_PRECISE=0; BASELINE config 2's `value` is the 32-sequence PRECISE run, the "
        "32-sequence plain run its companion `value_plain16_batch32`)": {
            "asserted": 3e-3, "measured": 2.0e-3, "test": "tests/test_fulldepth_gpu.py::test_llama_13b_40_layers_plain16_flow",
            "note": "40 layers; 7.5e-4 at 2 layers (tests/test_fullsize_gpu.py)"}},
    "bf16": {"rel_l2_vs_fp32_oracle": 1.2e-2, "after_50_unet_steps": 2.5e-2,
             "note": "bf16 eps = 7.8e-3: north_star's 1e-3 is not reachable with one bf16 plane per MFMA operand — the ViT / UNet operands are "
                     "single-plane. (The LLM's precise mode with two bf16 planes holds 8e-6 ... 1.2e-5 against the reference-executed "
                     "golden on 16-bit-checkpoint weights, tests/test_golden_gpu.py; 1.5e-3 was its figure on un-rounded fp32 fixture weights)"}}

CONFIGS = {
    0: "headline: 1x448px image in -> text + one 1024px image out (BASELINE configs 2+3 composed)",
    1: "de-tokenizer only: 1x448px ViT features (B=2 incl. zero image) -> ResamplerXLV2 -> ONE SDXL-UNet CFG-2 Euler step",
    2: "comprehension: 1x448px in (2 crops), 165-token prefill, 128-token greedy decode (text only)",
    3: "text->image: 64-token prompt -> 8 text tokens + <img>64</img> -> 50-step SDXL de-tokenize at 1024x1024",
    4: "edit: 1x448px in + 16-token instruction -> image block -> 50-step edit loop, UNet Bc=3 x 8-ch, gs 7.5 / igs 1.5",
    5: "any-res multi-image multi-turn: 4x896px (ViT B=20), ~1.5k-token prefill, 3 turns x 64 text tokens, last turn "
       "emits an image -> 50-step t2i de-tokenize",
}


class BenchTokenizer:
    """Stand-in for the LLaMA sentencepiece tokenizer (clm_llama_tokenizer_224loc_anyres): only the special-token ids
    matter for the hot path. <img>=32000, <img_00000..63>=32001..32064, </img>=32065, <patch>=32066, </patch>=32067."""
    eos_token_id = 2
    BOI, EOI, BOP, EOP = 32000, 32065, 32066, 32067

    def encode(self, s, add_special_tokens=False):
        import re
        out = []
        for tok in re.findall(r"<img_\d{5}>|<img>|</img>", s):
            out.append(32000 if tok == "<img>" else 32065 if tok == "</img>" else 32001 + int(tok[5:10]))
        return out

    def decode(self, ids, skip_special_tokens=False):
        return " ".join(str(int(i)) for i in ids)


USE_VAE = True   # set from --no-vae
VAE_PRECISION = "fp32"   # set from --vae-precision
BATCH = 16       # generations processed together per step on one GPU (set from --batch)
GRID_PINPOINTS = [[a * 448, b * 448] for a, b in ((1, 1), (1, 2), (1, 3), (2, 1), (3, 1), (1, 4), (4, 1), (2, 2))]


def build_models(dev, dtype, llm_comm=None, cfg_comm=None, need=("vit", "llm", "adapter"), edit=False, max_cache_len=1024,
                 unet_comm=None):
    """llm_comm / cfg_comm / unet_comm: tensor-parallel Llama, CFG-parallel loop and pixel-row-sharded UNet communicators
    (tools/bench_tp_latency.py only; the throughput bench leaves them None = one full replica per GPU). Returns
    (vit, agent, adapter); parts not in `need` are None."""
    from seedx_amd import synthetic as syn
    from seedx_amd.detokenizer import EulerDiscreteScheduler, ResamplerXLV2, SDXLAdapter, SDXLAdapterWithLatentImage
    from seedx_amd.llama import LlamaForCausalLM
    from seedx_amd.seed_x import ContinuousLVLM
    from seedx_amd.unet import SDXL_BASE_CONFIG, UNet2DConditionModel
    from seedx_amd.visual_encoder import Resampler, VisionTransformerWithAttnPool
    vit = agent = adapter = None
    if "vit" in need or "adapter" in need:                                      # the adapter needs the ViT for its negatives
        vit = VisionTransformerWithAttnPool(**syn.FULL_VIT)
        vit.load_state_dict(syn.vit_state_dict(syn.FULL_VIT, dev, dtype))
        vit.eval().to(dev, dtype=dtype)
        vit._pack()
    if "llm" in need:
        llm = LlamaForCausalLM(dict(syn.FULL_LLM), max_cache_len=max_cache_len, max_batch=BATCH, comm=llm_comm)
        llm.load_state_dict(syn.llama_state_dict(syn.FULL_LLM, dev, dtype))
        llm.to(dev, dtype)
        llm._pack()
        torch.cuda.empty_cache()
        H = syn.FULL_LLM["hidden_size"]
        agent = ContinuousLVLM(llm, Resampler(8, H, 32, kv_dim=4096), Resampler(8, 4096, 32, kv_dim=H), add_patch_pos=True,
                               vit_down=True)                                    # agent_seed_x_i.yaml
        agent.load_state_dict(syn.agent_state_dict(H, 4096, dev, dtype))
        agent.eval().to(dev, dtype)
    if "adapter" in need:
        ucfg = dict(SDXL_BASE_CONFIG, in_channels=8) if edit else dict(SDXL_BASE_CONFIG)
        unet = UNet2DConditionModel(comm=unet_comm, **ucfg)
        unet.load_state_dict(syn.unet_state_dict(unet.cfg, dev, dtype))
        res = ResamplerXLV2(normalize=False, **syn.FULL_XLV2)
        res.load_state_dict(syn.xlv2_state_dict(syn.FULL_XLV2, dev, dtype), prefix="resampler.")
        adapter = (SDXLAdapterWithLatentImage if edit else SDXLAdapter)(unet, res, vit_down=True)
        adapter.comm = cfg_comm
        vae = None
        if USE_VAE:
            from seedx_amd.vae import AutoencoderKL
            vae = AutoencoderKL()                                                    # SDXL vae/config.json defaults
            vae.load_state_dict(syn.vae_state_dict(vae, dev, dtype))
            vae.to(dev, dtype, precision=VAE_PRECISION)
            vae._pack()
        adapter.init_pipe(vae=vae, scheduler=EulerDiscreteScheduler(), visual_encoder=vit, image_transform=None,
                          dtype=dtype, device=dev)
        unet._pack()
        torch.cuda.empty_cache()
    return vit, agent, adapter


def make_inputs(dev, seed=0, size=448, n_images=1, extra_text=0):
    """(uint8 images resident in HBM, prompt ids, marker ids). Prompt layout of eval_img2text_seed_x_i.py:131-150:
    BOS [INST] {<patch>64</patch>}·(crops-1) <img>64</img> ~question tokens [/INST]\\n."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    images = [torch.randint(0, 256, (size, size, 3), generator=g, dtype=torch.uint8).to(dev) for _ in range(n_images)]
    crops = (size // 448) ** 2 + 1
    text = torch.randint(3, 32000, (400,), generator=g).tolist()
    T = BenchTokenizer
    ids = [1] + text[:7]
    for _ in range(n_images):
        for c in range(crops):
            last = c == crops - 1
            ids += [T.BOI if last else T.BOP] + [0] * 64 + [T.EOI if last else T.EOP]
    ids += text[7:7 + 25 + extra_text]
    return images, ids


def preprocess(images, dev):
    """any_res.py:158-201 + transforms.py:5-20 on the GPU: uint8 images → ([n_crops, 3, 448, 448] fp32, patch_pos)."""
    from seedx_amd import image_ops
    tf = image_ops.get_transform(type='clip', image_size=448, keep_ratio=False, device=dev)
    tens, pos = [], []
    for im in images:
        t, p = image_ops.process_anyres_image(im, tf, GRID_PINPOINTS, 448)
        tens.append(t)
        pos.append(p)
    return torch.cat(tens, dim=0), torch.cat(pos, dim=0)


def requests_for(vit, inp, dev, vit_comm=None):
    """Path A for BATCH requests (preprocessing + ViT on all BATCH·n_crops crops at once) → generate_batch requests.
    vit_comm (latency mode, tools/bench_tp_latency.py): the crops are split over the ranks and the features all-gathered."""
    from seedx_amd import image_ops
    images, ids = inp
    G = BATCH
    T = BenchTokenizer
    crops, ppos = preprocess(images, dev)                                      # identical image per request: preprocess once
    n = crops.shape[0]                                                         # per request … but run the ViT on every crop
    allc = crops if G == 1 else crops.repeat(G, 1, 1, 1)
    if vit_comm is not None and vit_comm.world > 1:
        from seedx_amd.parallel import split_batch_forward
        emb = split_batch_forward(vit, allc, vit_comm)
    else:
        emb = vit(allc)                                                        # [G·n, 256, 4096]
    ids_dev = torch.tensor(ids, dtype=torch.long, device=dev)
    mask = image_ops.marker_mask(ids_dev, T.BOI, T.EOI, T.BOP, T.EOP).view(1, -1)   # eval_img2text_seed_x_i.py:153-160
    return [dict(input_ids=[ids], image_embeds=emb[n * g:n * (g + 1)], embeds_cmp_mask=torch.tensor([True] * n),
                 ids_cmp_mask=mask, patch_positions=ppos) for g in range(G)]


def front_half(vit, agent, tok, inp, n_text, dev=None, vit_comm=None):
    """Paths A + B for BATCH requests: GPU preprocessing, ViT on all crops at once, then the BATCH greedy decodes in lock
    step. Returns the image-generation features [BATCH, 64, 4096]."""
    dev = dev or agent.llm.device
    reqs = requests_for(vit, inp, dev, vit_comm=vit_comm)
    outs = agent.generate_batch(tok, reqs, max_new_tokens=n_text + 66 + 1, eos_token_id=None,
                                force_image_at=n_text)                          # path B, G sequences in lock step
    for out in outs:
        assert out["has_img_output"] and out["img_gen_feat"].shape == (1, 64, 4096), "transcript did not yield an image"
    return torch.cat([o["img_gen_feat"] for o in outs], dim=0)


def back_half(adapter, feats, steps_unet, seed, **kw):
    """Path C for BATCH requests as one UNet batch of 2·BATCH (edit: 3·BATCH) CFG samples (enqueue-only: no host sync)."""
    G = feats.shape[0]
    out = adapter.generate(image_embeds=feats, num_inference_steps=steps_unet, seed=[seed * G + g for g in range(G)],
                           output_type="u8" if USE_VAE else "latent", **kw)
    assert out.shape == ((G, 1024, 1024, 3) if USE_VAE else (G, 4, 128, 128))
    return out


# ---------------------------------------------------------------------------------------------------------------------
# workloads (one per BASELINE config)
# ---------------------------------------------------------------------------------------------------------------------
class Workload:
    def __init__(self, a, dev, dtype):
        self.a, self.dev, self.dtype, self.tok = a, dev, dtype, BenchTokenizer()
        self.setup()

    def flops(self):
        raise NotImplementedError

    def describe(self):
        return CONFIGS[self.a.config]


class Headline(Workload):
    def setup(self):
        self.vit, self.agent, self.adapter = build_models(self.dev, self.dtype)
        self.inp = make_inputs(self.dev)
        assert len(self.inp[1]) == 165

    def step(self, seed):
        feats = front_half(self.vit, self.agent, self.tok, self.inp, self.a.text_tokens, self.dev)
        return back_half(self.adapter, feats, self.a.unet_steps, seed)

    def flops(self):
        return 2 * FLOP_VIT_CROP + (165 + 128) * FLOP_LLM_TOKEN + 2 * self.a.unet_steps * FLOP_UNET_SAMPLE \
            + (FLOP_VAE_DECODE if USE_VAE else 0.0)

    def describe(self):
        return ("SEED-X-I: 448x448 uint8 image -> GPU any-res/resize/normalise (2 crops) -> ViT-G -> 165-token prefill -> "
                "128 greedy tokens (%d text + <img> + 64 forced + </img>) -> %d-step SDXL-UNet CFG-2 de-tokenize @1024x1024 "
                "-> %s. Batch of %d generations in lock step. Per-model constant cached outside the step: the CFG negative "
                "branch's ViT(zeros) features (adapter_modules.py:110-116 recomputes them per call); the %d identical input "
                "images are resized once per step" % (self.a.text_tokens, self.a.unet_steps,
                           "SDXL VAE decode to a uint8 [1024,1024,3] image" if USE_VAE else "latents (VAE decode skipped)",
                           BATCH, BATCH))


class DetokOneStep(Workload):                                                   # config 1
    def setup(self):
        self.vit, _, self.adapter = build_models(self.dev, self.dtype, need=("vit", "adapter"))
        self.images, _ = make_inputs(self.dev)

    def step(self, seed):
        crops, _ = preprocess(self.images, self.dev)
        x = crops[-1:].repeat(BATCH, 1, 1, 1)                                   # the 448² global view of each request
        out = self.adapter.generate(image_tensor=x, num_inference_steps=1, seed=[seed * BATCH + g for g in range(BATCH)],
                                    output_type="latent")
        assert out.shape == (BATCH, 4, 128, 128)
        return out

    def flops(self):
        return 2 * FLOP_VIT_CROP + 2 * FLOP_UNET_SAMPLE


class Comprehension(Workload):                                                  # config 2
    def setup(self):
        self.vit, self.agent, _ = build_models(self.dev, self.dtype, need=("vit", "llm"))
        self.inp = make_inputs(self.dev)

    def step(self, seed):
        reqs = requests_for(self.vit, self.inp, self.dev)
        outs = self.agent.generate_batch(self.tok, reqs, max_new_tokens=128, eos_token_id=None)
        assert all(len(o["generate_ids"]) == 128 for o in outs)
        return outs

    def flops(self):
        return 2 * FLOP_VIT_CROP + (165 + 128) * FLOP_LLM_TOKEN


class TextToImage(Workload):                                                    # config 3
    def setup(self):
        _, self.agent, self.adapter = build_models(self.dev, self.dtype, need=("llm", "adapter"))
        g = torch.Generator().manual_seed(3)
        self.ids = [1] + torch.randint(3, 32000, (63,), generator=g).tolist()

    def step(self, seed):
        reqs = [dict(input_ids=[self.ids]) for _ in range(BATCH)]
        outs = self.agent.generate_batch(self.tok, reqs, max_new_tokens=8 + 66 + 1, eos_token_id=None, force_image_at=8)
        feats = torch.cat([o["img_gen_feat"] for o in outs], dim=0)
        return back_half(self.adapter, feats, self.a.unet_steps, seed)

    def flops(self):
        return (64 + 75) * FLOP_LLM_TOKEN + 2 * self.a.unet_steps * FLOP_UNET_SAMPLE + (FLOP_VAE_DECODE if USE_VAE else 0.0)


class Edit(Workload):                                                           # config 4
    def setup(self):
        self.vit, self.agent, self.adapter = build_models(self.dev, self.dtype, edit=True)
        self.inp = make_inputs(self.dev, extra_text=16)
        self.src = torch.randn(1, 4, 128, 128, generator=torch.Generator().manual_seed(7))   # VAE-encoded source stand-in

    def step(self, seed):
        feats = front_half(self.vit, self.agent, self.tok, self.inp, 8, self.dev)
        return back_half(self.adapter, feats, self.a.unet_steps, seed, image_latents=self.src.repeat(BATCH, 1, 1, 1))

    def flops(self):
        return 2 * FLOP_VIT_CROP + (181 + 75) * FLOP_LLM_TOKEN + 3 * self.a.unet_steps * FLOP_UNET_SAMPLE \
            + (FLOP_VAE_DECODE if USE_VAE else 0.0)


class MultiTurn(Workload):                                                      # config 5
    def setup(self):
        self.vit, self.agent, self.adapter = build_models(self.dev, self.dtype, max_cache_len=2304)
        self.inp = make_inputs(self.dev, size=896, n_images=4)
        g = torch.Generator().manual_seed(5)
        self.turn_text = [torch.randint(3, 32000, (20,), generator=g).tolist() for _ in range(2)]
        self.prefilled = None

    def step(self, seed):
        from seedx_amd import image_ops
        T = BenchTokenizer
        reqs = requests_for(self.vit, self.inp, self.dev)                       # ViT B = 20 per request
        self.agent._conv = {}                                                   # every step is a NEW conversation
        ids = list(self.inp[1])
        pre = []
        for turn in range(3):
            last = turn == 2
            for r in reqs:
                r["input_ids"] = [ids]
                r["ids_cmp_mask"] = image_ops.marker_mask(torch.tensor(ids, dtype=torch.long, device=self.dev), T.BOI,
                                                          T.EOI, T.BOP, T.EOP).view(1, -1)
            outs = self.agent.generate_batch(self.tok, reqs, max_new_tokens=(8 + 67) if last else 64, eos_token_id=None,
                                             force_image_at=8 if last else None, reuse_cache=bool(self.a.kv_reuse))
            pre.append(self.agent.last_prefill_tokens[0])
            if not last:
                ids = ids + outs[0]["generate_ids"].tolist() + self.turn_text[turn]   # every request shares the transcript
        self.prefilled = pre
        feats = torch.cat([o["img_gen_feat"] for o in outs], dim=0)
        return back_half(self.adapter, feats, self.a.unet_steps, seed)

    def flops(self):
        T0 = len(self.inp[1])
        toks = (T0 + 64) + (20 + 64) + (20 + 75) if self.a.kv_reuse else (T0 + 64) + (T0 + 84 + 64) + (T0 + 168 + 75)
        return 20 * FLOP_VIT_CROP + toks * FLOP_LLM_TOKEN + 2 * self.a.unet_steps * FLOP_UNET_SAMPLE \
            + (FLOP_VAE_DECODE if USE_VAE else 0.0)

    def describe(self):
        return CONFIGS[5] + "; cross-turn KV reuse %s (prefilled tokens per turn: %s; the reference re-prefills everything)" \
            % ("ON" if self.a.kv_reuse else "OFF", self.prefilled)


class StubWorkload(Workload):
    """CPU stand-in used by the launch-path test (tests/test_cpu_suite.py): exercises argument parsing, rank start-up,
    the barrier / max-over-ranks timing and the JSON line without a GPU or a model."""

    def setup(self):
        pass

    def step(self, seed):
        time.sleep(0.01)

    def flops(self):
        return 0.0


WORKLOADS = {0: Headline, 1: DetokOneStep, 2: Comprehension, 3: TextToImage, 4: Edit, 5: MultiTurn}


# ---------------------------------------------------------------------------------------------------------------------
class Pipeline:
    """Request-level software pipeline on two HIP streams (config 0 only): while the MFMA-bound de-tokenizer of request i
    runs on the `back` stream, the HBM-bound LLM decode (and ViT) of request i+1 runs on the `front` stream. Every request
    still executes all of its work; only the phases of CONSECUTIVE requests overlap."""

    def __init__(self, w):
        self.w = w
        self.s_front, self.s_back = torch.cuda.Stream(), torch.cuda.Stream()
        self.keep = []

    def submit(self, seed):
        w = self.w
        with torch.cuda.stream(self.s_front):
            feats = front_half(w.vit, w.agent, w.tok, w.inp, w.a.text_tokens, w.dev)
            ev = torch.cuda.Event()
            ev.record(self.s_front)
        with torch.cuda.stream(self.s_back):
            self.s_back.wait_event(ev)
            lat = back_half(w.adapter, feats, w.a.unet_steps, seed)
        self.keep.append((feats, lat))          # keep tensors alive until both streams are drained

    def drain(self):
        self.s_front.synchronize()
        self.s_back.synchronize()
        out = [l for _, l in self.keep]
        self.keep = []
        return out


PEAK_HBM_GBPS = 8000.0       # HBM3E peak (spec), MI355X_MICROARCH.md


def _kernel_source_sha():
    """Hash of the GEMM kernel sources: a stored PMC traffic profile is only quoted while these files are unchanged."""
    import hashlib
    h = hashlib.sha256()
    for f in ("gemm.hip", "gemm_pp.hip", "gemm_common.h", "sx_common.h"):
        with open(os.path.join(ROOT, "seed-x_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _csrc_sha():
    """Hash of every kernel / ABI source the line was measured with (the GPU box has no .git): ties a stored line to a tree."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "seed-x_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            with open(os.path.join(d, f), "rb") as fh:
                h.update(f.encode() + fh.read())
    with open(os.path.join(ROOT, "include", "seedx_hip.h"), "rb") as fh:
        h.update(fh.read())
    return h.hexdigest()[:16]


def measure_roofline(w):
    """Instrumented (eager, un-graphed, ONE kernel chain) step with HIP events on the launch stream around EVERY sx_gemm,
    sx_gemv and sx_attention launch, and around every phase of the step (ViT / prefill / decode / UNet loop / VAE):
      * per launch: algorithmic FLOPs (2·M·N·K, conv 2·M·N·9·Cin; K ÷ 3 for the fp32-grade VAE's plane-carrying launches;
        attention 4·Sq·Skv·D per head) and algorithmic bytes (operands once + output once), duration from the event pair
      * per phase: wall time between its two events, the FLOPs / bytes of the launches issued inside it
    Returns (roofline of the dominant kernel family, per-phase dict)."""
    from seedx_amd import _lib, ops
    lib = _lib.load()
    real = {n: getattr(lib, n) for n in ("sx_gemm", "sx_gemm_gn", "sx_gemm_ln", "sx_gemv", "sx_attention", "sx_attn_decode_b", "sx_attn_decode_fused",
                                         "sx_attention_f32")}
    rec = []                       # (family, phase, flops, bytes, start, end)
    executed = {}                  # phase -> MFMA FLOPs actually issued (plane-carrying launches counted with their tripled K)
    stack = ["other"]
    spans = {}                     # phase -> list of (start, end)

    def timed(fam, fl, byt, fn, *args):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = fn(*args)
        e.record()
        rec.append((fam, stack[-1], fl, byt, s, e))
        return r

    def h_gemm(args_ref, *rest):
        # sx_gemm(args, stream) | sx_gemm_gn(args, stats, groups, rows, fused, stream): the same launch, with the next GroupNorm's
        # statistics accumulated in its epilogue
        # | sx_gemm_ln(args, ln, stream): the LayerNorm ahead of / behind it folded in (a producer also writes the 16-bit copy)
        fn = real[{1: "sx_gemm", 2: "sx_gemm_ln"}.get(len(rest), "sx_gemm_gn")]
        a = args_ref._obj
        n_out = a.N // 2 if a.glu else a.N
        n_st = a.n_valid if a.n_valid else n_out
        pl = 2 if a.a_planes == 2 else 1                   # precise LLM: A carries the hi and lo planes (2x the A bytes, 2x the MFMA work)
        a_bytes = 2.0 * (a.B * a.Hin * a.Win * a.Cin if a.a_mode == 1 else a.M * a.K * pl)   # operands once + output once
        byt = a_bytes + 2.0 * a.N * a.K + a.M * n_st * (4.0 if a.out_dtype == 2 else 2.0) + (4.0 * a.M * n_st if a.residual else 0.0)
        if len(rest) == 2 and rest[0]._obj.row_stats_out:
            byt += 2.0 * a.M * a.N
        executed[stack[-1]] = executed.get(stack[-1], 0.0) + 2.0 * a.M * a.N * a.K * pl
        return timed("gemm", 2.0 * a.M * a.N * a.K / ops.OPERAND_PLANES, byt, fn, args_ref, *rest)

    def h_gemv(args_ref, stream):
        a = args_ref._obj
        n_out = a.N // 2 if a.glu else a.N
        pl = 2 if a.x_planes == 2 else 1
        byt = 2.0 * a.N * a.K + 2.0 * a.M * a.K * pl + a.M * n_out * (4.0 if (a.out_dtype & 0xff) == 2 else 2.0) + (4.0 * a.M * n_out if a.residual else 0.0)
        return timed("gemv", 2.0 * a.M * a.N * a.K, byt, real["sx_gemv"], args_ref, stream)

    def h_attn(args_ref, stream):
        a = args_ref._obj
        fl = 4.0 * a.B * a.H * a.Sq * a.Skv * a.D * (0.5 if a.causal else 1.0)
        byt = 2.0 * a.B * a.H * a.D * (2 * a.Sq + 2 * a.Skv)
        return timed("attention", fl, byt, real["sx_attention"], args_ref, stream)

    def h_attn_decode(*args):
        # split-KV decode attention: K and V of every visible key are read once (ctx lives on the device: read it back — this is
        # the instrumented eager pass, not the timed one)
        G, H, D = args[6], args[7], args[8]
        keys = float(w.agent.llm._P["ctx"][:G].sum().item())
        return timed("attn_decode", 4.0 * keys * H * D, 2.0 * 2.0 * keys * H * D, real["sx_attn_decode_b"], *args)

    def h_attn_decode_fused(args_ref, stream):
        a = args_ref._obj          # one-launch form: K and V of pos[g] + 1 keys per sequence
        keys = float(w.agent.llm._P["pos"][:a.G].sum().item()) + a.G
        return timed("attn_decode", 4.0 * keys * a.H * a.D, 2.0 * 2.0 * keys * a.H * a.D, real["sx_attn_decode_fused"], args_ref, stream)

    def h_attn_f32(args_ref, stream):
        # precise LLM: fp32 attention over the fp32 cache. T = 1 (decode step): K and V of pos[g] + 1 keys per sequence, 4 bytes each →
        # the decode phase's KV bytes; T > 1 (prefill / chunk): fp32 FMA work, its own family (not an MFMA kernel)
        a = args_ref._obj
        if not a.causal:
            keys, pairs = float(a.G) * a.Tmax, float(a.G) * a.T * a.Tmax
        else:
            pos = w.agent.llm._P["pos"][:a.G].float()
            keys = float((pos + a.T).sum().item())
            pairs = float((a.T * pos + a.T * (a.T + 1) / 2.0).sum().item())
        fam_ = "attn_decode" if (a.T == 1 and a.causal) else "attention_f32"
        return timed(fam_, 4.0 * pairs * a.H * a.D, 4.0 * 2.0 * keys * a.H * a.D, real["sx_attention_f32"], args_ref, stream)

    def phase_wrap(obj, name, phase):
        cls = type(obj)
        orig = getattr(cls, name)

        def wrapped(self_, *args, **kw):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            stack.append(phase)
            s.record()
            try:
                return orig(self_, *args, **kw)
            finally:
                e.record()
                stack.pop()
                spans.setdefault(phase, []).append((s, e))
        setattr(cls, name, wrapped)
        return cls, name, orig

    loop = getattr(getattr(w, "adapter", None), "_loop", None)
    agent = getattr(w, "agent", None)
    graphs = [m for m in (agent, loop) if m is not None]
    for m in graphs:
        m.use_graph = False
    chains = getattr(loop, "chains", 1)
    if loop is not None:
        loop.chains = 1          # ONE kernel chain on ONE stream: an event pair then brackets exactly its own launch (with the
                                 # two concurrent chains of the timed step it would also contain the other chain's kernels)
    patched = []
    try:
        if getattr(w, "vit", None) is not None:
            patched.append(phase_wrap(w.vit, "__call__", "vit"))
        if agent is not None:
            patched.append(phase_wrap(agent.llm, "forward_embeds_batch", "prefill"))
            patched.append(phase_wrap(agent.llm, "decode_step", "decode"))
        if loop is not None:
            patched.append(phase_wrap(loop, "run", "unet"))
            patched.append(phase_wrap(w.adapter, "_finish", "vae"))
        lib.sx_gemm, lib.sx_gemm_gn, lib.sx_gemm_ln, lib.sx_gemv, lib.sx_attention = h_gemm, h_gemm, h_gemm, h_gemv, h_attn
        if agent is not None:
            lib.sx_attn_decode_b = h_attn_decode
            lib.sx_attn_decode_fused = h_attn_decode_fused
            lib.sx_attention_f32 = h_attn_f32
        w.step(1)
        torch.cuda.synchronize()
    finally:
        for n, f in real.items():
            setattr(lib, n, f)
        for cls, name, orig in patched:
            setattr(cls, name, orig)
        for m in graphs:
            m.use_graph = True
        if loop is not None:
            loop.chains = chains

    fam = {}
    for f, ph, fl, byt, s, e in rec:
        d = fam.setdefault(f, {"n": 0, "s": 0.0, "flop": 0.0, "bytes": 0.0})
        d["n"] += 1; d["s"] += s.elapsed_time(e) * 1e-3; d["flop"] += fl; d["bytes"] += byt
    phases = {}
    for ph, sp in spans.items():
        wall = sum(s.elapsed_time(e) for s, e in sp) * 1e-3
        inside = [r_ for r_ in rec if r_[1] == ph]
        fl = sum(r_[2] for r_ in inside)
        byt_gemv = sum(r_[3] for r_ in inside if r_[0] == "gemv")
        t_fam = {f: sum(r_[4].elapsed_time(r_[5]) for r_ in inside if r_[0] == f) * 1e-3 for f in ("gemm", "gemv", "attention")}
        fl_fam = {f: sum(r_[2] for r_ in inside if r_[0] == f) for f in ("gemm", "gemv", "attention")}
        d = {"wall_ms": wall * 1e3, "calls": len(sp), "algorithmic_tflop": fl / 1e12,
             "tflops": fl / wall / 1e12 if wall > 0 else 0.0, "mfma_frac": fl / wall / 1e12 / PEAK_TFLOPS_16BIT if wall > 0 else 0.0,
             "kernel_time_share": {f: (t_fam[f] / wall if wall > 0 else 0.0) for f in t_fam},
             "family_tflops": {f: (fl_fam[f] / t_fam[f] / 1e12 if t_fam[f] > 0 else None) for f in t_fam}}
        ex = executed.get(ph, 0.0) + fl_fam["attention"]
        if ex > 1.001 * (fl_fam["gemm"] + fl_fam["attention"]) and wall > 0:
            # fp32-grade VAE: every fp32 product is three bf16 MFMA products (hi·hi + hi·lo + lo·hi); `frac` above prices the
            # ALGORITHMIC FLOPs, this is what the matrix pipe actually executes
            d["executed_mfma_tflop"] = ex / 1e12
            d["executed_mfma_frac"] = ex / wall / 1e12 / PEAK_TFLOPS_16BIT
        if ph == "decode":         # weight streaming: every skinny-GEMM launch reads its weight matrix once, decode attention the KV cache
            byt_kv = sum(r_[3] for r_ in inside if r_[0] == "attn_decode")
            t_kv = sum(r_[4].elapsed_time(r_[5]) for r_ in inside if r_[0] == "attn_decode") * 1e-3
            d.update({"bound": "hbm", "achieved": (byt_gemv + byt_kv) / wall / 1e9, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                      "frac": (byt_gemv + byt_kv) / wall / 1e9 / PEAK_HBM_GBPS,
                      "weight_bytes": byt_gemv, "kv_bytes": byt_kv,
                      "gemv_kernel_gbps": byt_gemv / t_fam["gemv"] / 1e9 if t_fam["gemv"] > 0 else None,
                      "attn_decode_kernel_gbps": byt_kv / t_kv / 1e9 if t_kv > 0 else None,
                      "kernel_time_share_attn_decode": t_kv / wall if wall > 0 else 0.0})
        else:
            d.update({"bound": "mfma", "achieved": d["tflops"], "peak": PEAK_TFLOPS_16BIT, "unit": "TFLOP/s", "frac": d["mfma_frac"]})
        phases[ph] = d
    if agent is not None and "decode" in phases and phases["decode"]["calls"] > 0:
        # the timed step REPLAYS the decode token step as a HIP graph; the eager pass above pays ~360 launches per token. Time
        # 32 graph replays at the position the step ended on (bytes per token from the eager pass: weights + visible KV).
        llm = agent.llm
        G = llm.G
        ids = torch.full((G, 40), -1, dtype=torch.int32, device=llm.device)
        hid = torch.zeros((G, 40, llm.config.hidden_size), device=llm.device)
        img_ids = torch.arange(llm.V - 200, llm.V - 134, dtype=torch.int32, device=llm.device)
        snap = {k: llm._P[k].clone() for k in ("pos", "ctx", "step", "cur")}
        llm._P["step"].zero_()
        llm._P["cur"].fill_(5)
        for _ in range(3):
            llm.decode_step(img_ids, ids, hid, use_graph=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(32):
            llm.decode_step(img_ids, ids, hid, use_graph=True)
        e1.record()
        torch.cuda.synchronize()
        for k, v in snap.items():
            llm._P[k].copy_(v)
        llm._graph = None
        ms_tok = e0.elapsed_time(e1) / 32
        dd = phases["decode"]
        byt_tok = (dd["weight_bytes"] + dd["kv_bytes"]) / dd["calls"]
        dd["graph_replay"] = {"ms_per_token": ms_tok, "bytes_per_token": byt_tok, "achieved": byt_tok / ms_tok / 1e6, "unit": "GB/s",
                              "frac": byt_tok / ms_tok / 1e6 / PEAK_HBM_GBPS,
                              "note": "32 HIP-graph replays of the token step for all sequences, as in the timed step; the eager figures "
                                      "above include ~360 host launches per token"}
    if "attention" in fam and fam["attention"]["s"] > 0:
        d = fam["attention"]
        phases["attention_kernels"] = {"bound": "mfma", "achieved": d["flop"] / d["s"] / 1e12, "peak": PEAK_TFLOPS_16BIT, "unit": "TFLOP/s",
                                       "frac": d["flop"] / d["s"] / 1e12 / PEAK_TFLOPS_16BIT, "launches": d["n"], "kernel_time_s": d["s"]}
    phases["_method"] = ("one eager step, one kernel chain; wall = HIP events around each phase on the launch stream; FLOPs / bytes = "
                         "algorithmic work of the sx_gemm / sx_gemv / sx_attention launches issued inside the phase")

    # dominant kernel family by summed launch time
    g, v = fam.get("gemm"), fam.get("gemv")
    if v is not None and (g is None or v["s"] > g["s"]):
        roof = {"bound": "hbm", "achieved": v["bytes"] / v["s"] / 1e9, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                "frac": v["bytes"] / v["s"] / 1e9 / PEAK_HBM_GBPS, "traffic": None, "traffic_unit": "bytes/launch",
                "traffic_source": "no PMC pass for this config", "algorithmic_bytes_per_launch": v["bytes"] / v["n"],
                "kernel": "sxk_decode::gemm_skinny_kernel<*> / gemv_kernel<*>", "launches_per_step": v["n"],
                "avg_launch_us": v["s"] / v["n"] * 1e6, "kernel_time_s_per_step": v["s"]}
        return roof, phases
    fl, tot_s, n, alg_bytes = g["flop"], g["s"], g["n"], g["bytes"]
    # traffic: bytes per launch from rocprofv3 --pmc passes of THIS command (tools/bench_pmc_traffic.py; eager launches).
    # Only quoted when the stored profile was taken with the GEMM sources as they are now, at this batch size / config.
    traffic, tnote, tscope = None, "no PMC profile taken with the current GEMM kernels at this batch size / dtype / config", None
    for name in ("r6_unet_pmc_traffic.json", "r5_unet_pmc_traffic.json"):
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", name)))
            if prof.get("batch_per_gpu") == BATCH and w.a.config == 0 and prof.get("kernel_source_sha") == _kernel_source_sha() \
                    and prof.get("dtype") == w.a.dtype:
                traffic, tnote = prof["traffic_bytes_per_launch"], prof["method"]
                tscope = {"scope": prof["scope"], "launches": prof["launches"],
                          "algorithmic_bytes_per_launch_same_launches": prof["algorithmic_bytes_per_launch"],
                          "traffic_over_algorithmic": prof["traffic_over_algorithmic"],
                          "attention_kernel": prof["families"].get("attention"),
                          "note": "PMC traffic and algorithmic bytes are over the SAME launches (the UNet's, 96 % of the step's GEMM time: "
                                  "rocprofv3 crashes with these counters on the composite bench command)"}
                break
        except (OSError, ValueError, KeyError):
            pass
    # ONE ratio readable off the line: `traffic` and `algorithmic_bytes_per_launch` are over the same launches whenever a PMC profile is
    # quoted; the figure over every sx_gemm launch of the step moves to its own, explicitly named key
    same = tscope["algorithmic_bytes_per_launch_same_launches"] if tscope else None
    roof = {"bound": "mfma", "achieved": fl / tot_s / 1e12, "peak": PEAK_TFLOPS_16BIT, "unit": "TFLOP/s",
            "frac": fl / tot_s / 1e12 / PEAK_TFLOPS_16BIT, "traffic": traffic, "traffic_unit": "bytes/launch",
            "traffic_source": tnote, "traffic_detail": tscope,
            "algorithmic_bytes_per_launch": same if same is not None else alg_bytes / n,
            "traffic_over_algorithmic": (traffic / same) if (traffic and same) else None,
            "algorithmic_bytes_per_launch_all_gemm_launches_of_the_step": alg_bytes / n,
            "kernel": "sxk_gemm::gemm_pp_kernel<*> + gemm_kernel<*> (every sx_gemm launch)",
            "launches_per_step": n, "avg_launch_us": tot_s / n * 1e6, "avg_launch_gflop": fl / n / 1e9,
            "gemm_time_s_per_step": tot_s}
    return roof, phases


def _median_time(fn, reps=3, warm=1):
    """SURVEY.md §8(d) protocol: `warm` untimed calls, then the median of `reps` timed calls."""
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.time()
        fn()
        ts.append(time.time() - t0)
    ts.sort()
    return ts[len(ts) // 2]


def cpu_baseline(config=0, full_config1=False):
    """CPU oracle ("port": the functions of oracle/restated*.py, fp32 torch on the host's PHYSICAL cores), 1 warm-up + 3
    timed repetitions (median) of every sampled piece, extrapolated by layer count / algorithmic FLOPs to one config-0
    generation. `full_config1` (bench.py --config 1 --cpu-baseline full): BASELINE config 1 end to end, un-extrapolated
    (full ViT-G forward of 2 crops + ResamplerXLV2 + ONE full SDXL-UNet CFG-2 step), one timed run after a warm-up block."""
    from oracle import restated, restated_unet as ru, weights
    try:
        import psutil
        cores = psutil.cpu_count(logical=False) or os.cpu_count()
    except Exception:
        cores = os.cpu_count()
    # thread count: the fastest of {16, 32, 64, all physical cores} on one ViT block (fp32 GEMMs of this size stop scaling
    # long before 128 threads; the reported `cores` is the count actually used)
    g = torch.Generator().manual_seed(0)
    cfg1 = dict(weights.FULL_VIT, layers=1)
    sd1 = weights.vit_sd(cfg1)
    x1 = torch.randn(1, 3, 448, 448, generator=g)
    best = (1e30, max(1, cores))
    for nt_ in sorted({min(c_, max(1, cores)) for c_ in (16, 32, 64, cores)}):
        torch.set_num_threads(nt_)
        t_ = _median_time(lambda: restated.vit_forward(sd1, cfg1, x1), reps=2)
        if t_ < best[0]:
            best = (t_, nt_)
    del sd1
    torch.set_num_threads(best[1])
    cores = torch.get_num_threads()
    if full_config1:
        cfg = dict(weights.FULL_VIT)
        sd = weights.vit_sd(cfg)
        x = torch.randn(2, 3, 448, 448, generator=g)
        restated.vit_forward(sd, dict(cfg, layers=1), x[:1])                      # warm-up (allocator, thread pool)
        t0 = time.time()
        restated.vit_forward(sd, cfg, x)
        t_vit = time.time() - t0
        del sd
        ucfg = ru.FULL_UNET
        usd = {k: torch.randn(s_, generator=g) * (0.02 if len(s_) > 1 else 1.0) for k, s_ in ru.unet_param_shapes(ucfg).items()}
        lat = torch.randn(2, 4, 128, 128, generator=g)
        ehs = torch.randn(2, 64, 2048, generator=g)
        t0 = time.time()
        ru.unet_forward(usd, ucfg, lat, torch.tensor(999.0), ehs, torch.randn(2, 1280, generator=g),
                        torch.tensor([[1024., 1024., 0., 0., 1024., 1024.]] * 2))
        t_unet = time.time() - t0
        t_total = t_vit + t_unet
        return {"value": 1.0 / t_total, "unit": "gens/s", "cores": cores, "kind": "port",
                "sample": "BASELINE config 1 un-extrapolated on %d host threads: oracle ViT-G forward of 2 crops %.1fs + ONE full "
                          "SDXL-UNet CFG-2 step at 128x128 latents %.1fs (single timed run each after a thread-pool warm-up; "
                          "ResamplerXLV2 < 0.1 s omitted)" % (cores, t_vit, t_unet)}
    # (1) oracle ViT (restated.vit_forward) at full width with 1 and 2 of the 48 blocks, B = 2 crops: the difference is one
    #     block, the 1-block run carries patchify + attn_pool + proj
    x = torch.randn(2, 3, 448, 448, generator=g)
    t_vit = []
    for layers in (1, 2):
        cfg = dict(weights.FULL_VIT, layers=layers)
        sd = weights.vit_sd(cfg)
        t_vit.append(_median_time(lambda: restated.vit_forward(sd, cfg, x)))
        del sd
    t_block = max(t_vit[1] - t_vit[0], 1e-3)
    t_total = t_vit[0] + 47 * t_block
    # (2) oracle Llama (restated.llama_forward), one of 40 layers: prefill T=165 and cached decode steps
    cfg = dict(weights.FULL_LLM, num_hidden_layers=1, vocab_size=512)
    sd = weights.llama_sd(cfg)
    xe = torch.randn(1, 165, 5120, generator=g)
    t_pre = _median_time(lambda: restated.llama_forward(sd, cfg, xe))
    _, past0, _ = restated.llama_forward(sd, cfg, xe)

    def dec4():
        past = past0
        for _ in range(4):
            _, past, _ = restated.llama_forward(sd, cfg, xe[:, :1], past)
    t_dec = _median_time(dec4) / 4
    n_new = 128
    t_total += 40 * (t_pre + n_new * t_dec)
    # (3) oracle SDXL UNet pieces (restated_unet._resnet / _transformer): mid-block resnet + one of its 10 transformer
    #     layers at 32x32, CFG batch 2 → scaled by algorithmic FLOPs to the whole UNet
    ucfg = ru.FULL_UNET
    shapes = ru.unet_param_shapes(ucfg)
    usd = {}
    for k, s_ in shapes.items():
        if k.startswith("mid_block.resnets.0") or k.startswith("mid_block.attentions.0.norm") or \
                k.startswith("mid_block.attentions.0.proj") or k.startswith("mid_block.attentions.0.transformer_blocks.0."):
            usd[k] = torch.randn(s_, generator=g) * (0.02 if len(s_) > 1 else 1.0)
    xs = torch.randn(2, 1280, 32, 32, generator=g)
    emb = torch.randn(2, 1280, generator=g)
    ehs = torch.randn(2, 64, 2048, generator=g)

    def unet_piece():
        h = ru._resnet(usd, "mid_block.resnets.0", xs, emb, 32)
        ru._transformer(usd, "mid_block.attentions.0", h, ehs, 20, 1, 32)
    t_u = _median_time(unet_piece)
    C, HW, B = 1280, 1024, 2
    fl_sample = B * (2 * 2 * HW * C * 9 * C + 2 * HW * C * C * 2            # resnet convs + proj_in/out
                     + 2 * HW * C * (3 * C + C + C + C + 8 * C + 4 * C)        # qkv, out, q2, out2, geglu, ff2
                     + 4 * HW * HW * C + 4 * HW * 64 * C + 2 * 2 * 64 * 2048 * C)
    t_total += 50 * t_u * (2 * FLOP_UNET_SAMPLE / fl_sample)
    return {"value": 1.0 / t_total, "unit": "gens/s", "cores": cores, "kind": "port",
            "sample": "oracle/restated*.py fp32 torch on %d host threads (physical cores), 1 warm-up + median of 3 runs per piece: "
                      "ViT-G with 1 and 2 of 48 blocks (B=2) %.2fs / %.2fs, 1 of 40 Llama-13B-dim layers (prefill T=165 %.2fs, cached "
                      "decode %.3fs/token), SDXL mid-block resnet + 1 transformer layer @32x32 CFG-2 %.2fs; extrapolated by layer "
                      "count / algorithmic FLOPs to one config-0 generation (%.0f s)"
                      % (cores, t_vit[0], t_vit[1], t_pre, t_dec, t_u, t_total)}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=0, choices=sorted(CONFIGS), help="; ".join(f"{k}: {v}" for k, v in CONFIGS.items()))
    ap.add_argument("--dtype", default="fp16", choices=["bf16", "fp16"],
                    help="operand dtype of the timed run. Default fp16 = the reference scripts' dtype (eval_*.py) and the one whose "
                         "full-size parity is asserted at north_star's 1e-3 (bf16: 1.2e-2, DESIGN.md §7)")
    ap.add_argument("--also-dtype", default="auto", choices=["auto", "none", "bf16", "fp16"],
                    help="after the timed run, time a short pass (1 warm-up + 2 steps) in a second dtype and report it as "
                         "value_<dtype> in the same JSON line. auto: the other 16-bit type for config 0 on one GPU, else none")
    ap.add_argument("--unet-steps", type=int, default=50)
    ap.add_argument("--text-tokens", type=int, default=61)
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of HIP-graph replay (profiling aid)")
    ap.add_argument("--overlap", type=int, default=0,
                    help="1: pipeline consecutive requests on two streams (LLM decode of request i+1 under the UNet of request i)")
    ap.add_argument("--batch", type=int, default=None,
                    help="independent generations processed together per GPU per step (default 16; config 5: 4; 1 = latency mode)")
    ap.add_argument("--kv-reuse", type=int, default=1, help="config 5: keep the KV cache across turns (0 = re-prefill like the reference)")
    ap.add_argument("--no-companion", action="store_true", help="config 2: skip the 32-sequence plain-flow companion pass")
    ap.add_argument("--chains", type=int, default=2, help="concurrent UNet kernel chains per denoise step (1 = one serial chain; "
                    "use 1 under rocprofv3 so per-kernel durations do not contain the other chain)")
    ap.add_argument("--no-vae", action="store_true", help="stop at the denoised latents (no VAE decode)")
    ap.add_argument("--vae-precision", default="fp32", choices=["fp32", "fast", "auto"],
                    help="fp32 (default): the reference's scripts decode with the VAE upcast to fp32 — operands carried as two "
                         "bf16 planes, fp32 accumulation; fast: single 16-bit operands; auto: what the reference would do for --dtype")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", default="sample", choices=["sample", "full"],
                    help="sample (default): bounded pieces, 1 warm-up + median of 3, extrapolated to a config-0 generation; full (with "
                         "--config 1): BASELINE config 1 end to end on the host, un-extrapolated (minutes)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--profile-markers", action="store_true",
                    help="bracket the timed region with sx_profile_marker dispatches (tools/kstats_step.py: per-kernel table of the step only)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo: CPU launch-path test only (with --stub)")
    ap.add_argument("--stub", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args(argv)


def main(argv=None):
    a = parse_args(argv)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no external launcher: start the N ranks ourselves (one process per GPU over RCCL) and hand over
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr",
               "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv)
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.run(cmd, env=env).returncode)
    global BATCH, USE_VAE, VAE_PRECISION
    # config 2 (text only): 32 lock-step sequences in the LLM's precise mode (round 6: four operand blocks per weight fragment), whose 40-layer
    # logits are asserted at north_star's 1e-3; the 32-sequence plain 16-bit flow (faster, 2.0e-3) runs as a companion pass into
    # `value_plain16_batch32`
    BATCH = a.batch if a.batch is not None else {5: 4, 2: 32}.get(a.config, 16)
    a.batch = BATCH
    USE_VAE = not a.no_vae
    VAE_PRECISION = a.vae_precision
    from seedx_amd import dist_utils as du
    ctx = du.init(a.backend)                    # RCCL over xGMI; only barrier + max-reduce of the wall time
    rank, world, local = ctx.rank, ctx.world, ctx.local
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but {world} rank(s) came up (WORLD_SIZE={os.environ.get('WORLD_SIZE')})")
    gpu = not a.stub
    if gpu:
        torch.cuda.set_device(local)
    dev = torch.device("cuda", local) if gpu else torch.device("cpu")
    seen = du.ranks_seen(ctx, strict_devices=bool(os.environ.get("SX_BENCH_STRICT_DEVICES")))   # every rank, its device: in the JSON line
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    sync = torch.cuda.synchronize if gpu else (lambda: None)
    with torch.no_grad():
        w = (StubWorkload if a.stub else WORKLOADS[a.config])(a, dev, dtype)
        if gpu and getattr(getattr(w, "adapter", None), "_loop", None) is not None:
            w.adapter._loop.chains = a.chains
        if a.no_graph and gpu:
            for m in (getattr(w, "agent", None), getattr(getattr(w, "adapter", None), "_loop", None)):
                if m is not None:
                    m.use_graph = False
        seeds = du.shard_seeds(ctx, a.steps)      # independent requests, round-robin over ranks
        pipe = Pipeline(w) if (a.overlap and a.config == 0 and gpu) else None

        def run(seed_list):
            if pipe is None:
                for sd_ in seed_list:
                    w.step(sd_)
            else:
                for sd_ in seed_list:
                    pipe.submit(sd_)
                pipe.drain()
        run([100 + i for i in range(a.warmup)])
        sync()
        du.barrier(ctx)
        sync()
        if a.profile_markers and gpu:
            from seedx_amd import _lib as _sxlib
            _sxlib.load().sx_profile_marker(1, torch.cuda.current_stream().cuda_stream)
            sync()
        t0 = time.perf_counter()
        run(seeds)
        sync()
        if a.profile_markers and gpu:
            _sxlib.load().sx_profile_marker(2, torch.cuda.current_stream().cuda_stream)
            sync()
        du.barrier(ctx)
        sync()
        dt_local = time.perf_counter() - t0
        dt = du.max_over_ranks(ctx, dt_local)
        # per-rank line on stderr (the driver's log shows every rank's own clock next to the max that the JSON line uses)
        print(f"[bench rank {rank}/{world}] device {seen[rank].get('device')} {seen[rank].get('device_id')}: {a.steps} steps x {a.batch} "
              f"generations in {dt_local:.3f} s (max over ranks {dt:.3f} s)", file=sys.stderr, flush=True)
        # what actually ran (the model's own flag, not the command line): ADVICE r5
        _llm = getattr(getattr(w, "agent", None), "llm", None)
        llm_mode = None if _llm is None else (
            ("precise (fp32-grade activations: two 16-bit operand planes, fp32 q / k / RoPE / attention; KV cache: k fp32, v %s; 40-layer "
             "logits asserted at 1e-3)" % ("16-bit (mixed cache)" if getattr(_llm, "kv_v16", False) else "fp32"))
            if _llm.precise else "plain 16-bit (one rounding per MFMA operand, 16-bit KV cache; 40-layer logits asserted at 3e-3)")
        roof = phases = None
        if rank == 0 and gpu and not a.no_roofline:
            try:
                roof, phases = measure_roofline(w)
            except Exception as ex:      # the timed result above stands on its own: report the failure instead of losing the line
                roof, phases = {"error": repr(ex)}, None
        # second dtype, short pass (same workload, same kernels, other operand type) → value_<dtype> in the same line
        other = a.also_dtype
        if other == "auto":
            other = ({"fp16": "bf16", "bf16": "fp16"}[a.dtype]) if (a.config == 0 and world == 1 and gpu) else "none"
        second = None
        if other not in ("none", a.dtype) and gpu and world == 1:
            try:
                import gc
                desc0, flops0 = w.describe(), w.flops()
                w = pipe = None
                gc.collect()
                torch.cuda.empty_cache()
                w2 = WORKLOADS[a.config](a, dev, torch.bfloat16 if other == "bf16" else torch.float16)
                if getattr(getattr(w2, "adapter", None), "_loop", None) is not None:
                    w2.adapter._loop.chains = a.chains
                w2.step(100)
                sync()
                t1 = time.perf_counter()
                for sd_ in (0, 1):
                    w2.step(sd_)
                sync()
                dt2 = time.perf_counter() - t1
                second = {"dtype": other, "value": 2 * a.batch / dt2, "ms_per_step": dt2 / 2 * 1e3, "steps": 2, "warmup": 1}
                w = w2
            except Exception as ex:
                second = {"dtype": other, "error": repr(ex)}
        # BASELINE config 2 companion: the same workload at 32 lock-step sequences = the plain 16-bit LLM flow (outside north_star's
        # 1e-3: 2.3e-3 at 40 layers, asserted at 3e-3 by tests/test_fulldepth_gpu.py::test_llama_13b_40_layers_plain16_flow)
        plain32 = None
        if a.config == 2 and gpu and world == 1 and not a.no_companion:
            try:
                import gc
                desc2, flops2 = w.describe(), w.flops()
                w = pipe = None
                gc.collect()
                torch.cuda.empty_cache()
                BATCH = 32
                os.environ["SX_LLM_PRECISE"] = "0"
                try:
                    w3 = WORKLOADS[2](a, dev, dtype)
                finally:
                    os.environ.pop("SX_LLM_PRECISE", None)
                assert not w3.agent.llm.precise
                w3.step(100)
                sync()
                t1 = time.perf_counter()
                for sd_ in (0, 1, 2):
                    w3.step(sd_)
                sync()
                dt3 = time.perf_counter() - t1
                plain32 = {"value": 3 * 32 / dt3, "unit": "gens/s", "batch_per_gpu": 32, "ms_per_step": dt3 / 3 * 1e3, "steps": 3, "warmup": 1,
                           "llm_mode": "plain 16-bit", "parity_bound": {"asserted": 3e-3, "measured": 2.0e-3,
                           "test": "tests/test_fulldepth_gpu.py::test_llama_13b_40_layers_plain16_flow"}}
                w3 = None
                BATCH = a.batch
            except Exception as ex:
                plain32 = {"error": repr(ex)}
                BATCH = a.batch
    if rank == 0:
        total = du.total_units(ctx, a.steps) * a.batch
        rec = {"metric": "end-to-end generations/sec (img-in -> txt + 1024px-img-out)" if a.config == 0 else
               "generations/sec of BASELINE config %d" % a.config, "value": total / dt,
               "unit": "gens/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ranks_seen": [{k: g.get(k) for k in ("rank", "local_rank", "host", "device", "device_id", "shared_device") if k in g}
                              for g in seen],
               "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": a.dtype if gpu else "none",
               "data": "synthetic (seeded random uint8 image / prompt ids, random-init weights of the real dims)",
               "config": {"workload": (desc0 if second is not None else (w.describe() if w is not None else None)) if gpu else "stub (launch-path test)",
                          "baseline_config": a.config,
                          "llm_mode": llm_mode if gpu else None,
                          "parity_bound": PARITY_BOUND.get(a.dtype) if gpu else None,
                          "parallelism": "replica x%d (independent generations, no data-path collective)" % world,
                          "batch_per_gpu": a.batch, "request_pipelining": bool(a.overlap),
                          "vae": ("none" if not USE_VAE else {"fp32": "fp32-grade (fp16: two fp16 activation planes x exact fp16 weights; bf16: three bf16 plane products; fp32 accumulation)",
                                                              "fast": "single 16-bit operands"}.get(VAE_PRECISION, "auto"))},
               "flops_per_generation": flops0 if second is not None else (w.flops() if w is not None else None), "generations_per_step": a.batch}
        if gpu and a.config == 2 and plain32 is not None:
            rec["value_plain16_batch32"] = plain32.get("value")
            rec["plain16_batch32"] = plain32
            if w is None:
                rec["config"]["workload"], rec["flops_per_generation"] = desc2, flops2
        if second is not None:
            if "error" in second:
                rec["value_" + second["dtype"]] = None
                rec["second_dtype"] = second
            else:
                rec["value_" + second["dtype"]] = second["value"]
                rec["second_dtype"] = dict(second, parity_bound=PARITY_BOUND.get(second["dtype"]),
                                           note="same workload and kernels in the other 16-bit operand type; short pass after the timed region")
        rec["model_tflops"] = rec["flops_per_generation"] * rec["value"] / world / 1e12
        rec["source_sha"] = {"csrc": _csrc_sha(), "gemm": _kernel_source_sha()}
        if roof is not None:
            rec["roofline"] = roof
            rec["roofline_phases"] = phases
        if gpu and not a.no_cpu_baseline and world == 1:
            try:
                rec["cpu_baseline"] = cpu_baseline(a.config, full_config1=(a.cpu_baseline == "full" and a.config == 1))
                # the un-extrapolated CPU figure (BASELINE config 1 end to end on the host cores) is measured once per
                # round by `bench.py --config 1 --cpu-baseline full` (≈ 1-2 min of CPU) and quoted here from its committed line
                f1 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r6_bench_config1_cpu_full.json")
                if not os.path.exists(f1):
                    f1 = f1.replace("r6_", "r5_")
                if a.config == 0 and "error" not in rec["cpu_baseline"] and os.path.exists(f1):
                    c1 = json.load(open(f1))
                    cb = c1.get("cpu_baseline", {})
                    if "value" in cb:
                        rec["cpu_baseline"]["config1_unextrapolated"] = {
                            "cpu_value": cb["value"], "unit": cb.get("unit"), "cores": cb.get("cores"), "gpu_value": c1.get("value"),
                            "sample": cb.get("sample"), "file": "profiles/" + os.path.basename(f1)}
                        rec["cpu_baseline"]["sample"] += ("; un-extrapolated companion (BASELINE config 1, ViT-G forward of 2 crops + ONE "
                                                          "UNet CFG-2 step, %s cores): CPU %.4g vs GPU %.4g gens/s"
                                                          % (cb.get("cores"), cb["value"], c1.get("value") or float("nan")))
            except Exception as ex:  # the oracle is optional infrastructure; never fail the measurement on it
                rec["cpu_baseline"] = {"error": repr(ex)}
        print(json.dumps(rec), flush=True)
    du.barrier(ctx)                 # rank 0 ran the instrumented roofline pass alone: tear the group down together
    if world > 1 and gpu:
        torch.cuda.synchronize()
    du.finalize(ctx)


if __name__ == "__main__":
    main()
