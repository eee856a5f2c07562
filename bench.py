#!/usr/bin/env python
"""bench.py — end-to-end SEED-X-I generations/sec on MI355X (BASELINE.json metric).

One "step" = ONE full generation on one GPU, with every input already resident in HBM:
    1x448px image (any-res → 2 crops) → ViT-G/448 → input resampler → Llama-13B-dim prefill (165 tokens) →
    greedy decode of 128 new tokens (61 text tokens with lm_head + logits rule, then <img> + 64 forced image tokens
    + </img>; EOS disabled so the length is fixed) → output resampler → ResamplerXLV2 (CFG batch 2; the all-zero-image
    negative ViT features are a per-model constant and cached) → 50-step SDXL UNet CFG(7.5)+Euler at 128x128 latents
    (= 1024x1024 px) → SDXL VAE decoder → [3, 1024, 1024] image in [0, 1] (`--no-vae` stops at the latents).
Synthetic data: seeded random image / prompt ids, random-init weights of the real architecture (no checkpoints exist
here). N > 1 GPUs: one process per GPU (torchrun), independent generations per rank (no data-path collective; weak
scaling); value = total generations of all ranks / max-over-ranks wall time.

Usage: python bench.py --gpus N --steps K --warmup W   (prints ONE JSON line on rank 0)
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_TFLOPS_16BIT = 2500.0   # dense bf16/fp16 MFMA peak, MI355X_MICROARCH.md
# algorithmic FLOPs
FLOP_VIT_CROP = 4.219e12
FLOP_LLM_TOKEN = 25.71e9
FLOP_UNET_SAMPLE = 6.747e12
FLOP_VAE_DECODE = 10.47e12   # SDXL VAE decoder at 128x128 latents (conv/linear/attention MACs x 2)


class BenchTokenizer:
    """Stand-in for the LLaMA sentencepiece tokenizer (clm_llama_tokenizer_224loc_anyres): only the special-token ids
    matter for the hot path. <img>=32000, <img_00000..63>=32001..32064, </img>=32065 (ids of the added tokens)."""
    eos_token_id = 2

    def encode(self, s, add_special_tokens=False):
        import re
        out = []
        for tok in re.findall(r"<img_\d{5}>|<img>|</img>", s):
            out.append(32000 if tok == "<img>" else 32065 if tok == "</img>" else 32001 + int(tok[5:10]))
        return out

    def decode(self, ids, skip_special_tokens=False):
        return " ".join(str(int(i)) for i in ids)


USE_VAE = True   # set from --no-vae


def build_models(dev, dtype, llm_comm=None, cfg_comm=None):
    """llm_comm / cfg_comm: tensor-parallel Llama and CFG-parallel UNet communicators (tools/bench_tp_latency.py only;
    the throughput bench leaves them None = one full replica per GPU)."""
    from seedx_amd import synthetic as syn
    from seedx_amd.detokenizer import EulerDiscreteScheduler, ResamplerXLV2, SDXLAdapter
    from seedx_amd.llama import LlamaForCausalLM
    from seedx_amd.seed_x import ContinuousLVLM
    from seedx_amd.unet import SDXL_BASE_CONFIG, UNet2DConditionModel
    from seedx_amd.visual_encoder import Resampler, VisionTransformerWithAttnPool
    vit = VisionTransformerWithAttnPool(**syn.FULL_VIT)
    vit.load_state_dict(syn.vit_state_dict(syn.FULL_VIT, dev, dtype))
    vit.eval().to(dev, dtype=dtype)
    vit._pack()
    llm = LlamaForCausalLM(dict(syn.FULL_LLM), max_cache_len=1024, max_batch=BATCH, comm=llm_comm)
    llm.load_state_dict(syn.llama_state_dict(syn.FULL_LLM, dev, dtype))
    llm.to(dev, dtype)
    llm._pack()
    torch.cuda.empty_cache()
    H = syn.FULL_LLM["hidden_size"]
    agent = ContinuousLVLM(llm, Resampler(8, H, 32, kv_dim=4096), Resampler(8, 4096, 32, kv_dim=H), add_patch_pos=True,
                           vit_down=True)                                    # agent_seed_x_i.yaml
    agent.load_state_dict(syn.agent_state_dict(H, 4096, dev, dtype))
    agent.eval().to(dev, dtype)
    unet = UNet2DConditionModel(**SDXL_BASE_CONFIG)
    unet.load_state_dict(syn.unet_state_dict(unet.cfg, dev, dtype))
    res = ResamplerXLV2(normalize=False, **syn.FULL_XLV2)
    res.load_state_dict(syn.xlv2_state_dict(syn.FULL_XLV2, dev, dtype), prefix="resampler.")
    adapter = SDXLAdapter(unet, res, vit_down=True)
    adapter.comm = cfg_comm
    vae = None
    if USE_VAE:
        from seedx_amd.vae import AutoencoderKL
        vae = AutoencoderKL()                                                    # SDXL vae/config.json defaults
        vae.load_state_dict(syn.vae_state_dict(vae, dev, dtype))
        vae.to(dev, dtype)
        vae._pack()
    adapter.init_pipe(vae=vae, scheduler=EulerDiscreteScheduler(), visual_encoder=vit, image_transform=None,
                      discrete_model=None, dtype=dtype, device=dev)
    unet._pack()
    torch.cuda.empty_cache()
    return vit, agent, adapter


def make_inputs(dev, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    image = torch.randn(2, 3, 448, 448, generator=g).to(dev)                   # 1 tile + global crop (any_res.py:185-189)
    patch_pos = torch.tensor([[0.5, 0.5], [0.5, 0.5]])
    # prompt: BOS [INST] <patch>64</patch> <img>64</img> ~20 question tokens [/INST]\n  → 165 tokens (SURVEY §8d cfg 2)
    text = torch.randint(3, 32000, (165,), generator=g).tolist()
    ids = [1] + text[:7] + [32066] + [0] * 64 + [32067] + [32000] + [0] * 64 + [32065] + text[7:7 + 26]
    ids = ids[:165]
    mask = torch.zeros(1, len(ids), dtype=torch.bool)
    mask[0, 9:73] = True
    mask[0, 75:139] = True
    return image, patch_pos, ids, mask


BATCH = 16  # generations processed together per step on one GPU (set from --batch)


def front_half(vit, agent, tok, inp, n_text):
    """Paths A + B for BATCH requests: ViT on all 2·BATCH crops at once, then the BATCH greedy decodes in lock step.
    Returns the image-generation features [BATCH, 64, 4096]."""
    image, patch_pos, ids, mask = inp
    G = BATCH
    emb = vit(image if G == 1 else image.repeat(G, 1, 1, 1))                    # path A, [2G, 256, 4096]
    reqs = [dict(input_ids=[ids], image_embeds=emb[2 * g:2 * g + 2], embeds_cmp_mask=torch.tensor([True, True]),
                 ids_cmp_mask=mask, patch_positions=patch_pos) for g in range(G)]
    outs = agent.generate_batch(tok, reqs, max_new_tokens=n_text + 66 + 1, eos_token_id=None,
                                force_image_at=n_text)                          # path B, G sequences in lock step
    for out in outs:
        assert out["has_img_output"] and out["img_gen_feat"].shape == (1, 64, 4096), "transcript did not yield an image"
    return torch.cat([o["img_gen_feat"] for o in outs], dim=0)


def back_half(adapter, feats, steps_unet, seed):
    """Path C for BATCH requests as one UNet batch of 2·BATCH CFG samples (enqueue-only: no host sync inside)."""
    G = feats.shape[0]
    out = adapter.generate(image_embeds=feats, num_inference_steps=steps_unet, seed=[seed * G + g for g in range(G)],
                           output_type="pt" if USE_VAE else "latent")
    assert out.shape == ((G, 3, 1024, 1024) if USE_VAE else (G, 4, 128, 128))
    return out


def one_generation(vit, agent, adapter, tok, inp, steps_unet, n_text, seed):
    """One bench step, strictly sequential on the current stream."""
    feats = front_half(vit, agent, tok, inp, n_text)
    return feats, back_half(adapter, feats, steps_unet, seed)


class Pipeline:
    """Request-level software pipeline on two HIP streams: while the MFMA-bound de-tokenizer of request i runs on the
    `back` stream, the HBM-bound LLM decode (and ViT) of request i+1 runs on the `front` stream. Every request still
    executes all of its work; only the phases of CONSECUTIVE requests overlap."""

    def __init__(self, vit, agent, adapter, tok, inp, steps_unet, n_text):
        self.args = (vit, agent, adapter, tok, inp, steps_unet, n_text)
        self.s_front, self.s_back = torch.cuda.Stream(), torch.cuda.Stream()
        self.keep = []

    def submit(self, seed):
        vit, agent, adapter, tok, inp, steps_unet, n_text = self.args
        with torch.cuda.stream(self.s_front):
            feats = front_half(vit, agent, tok, inp, n_text)
            ev = torch.cuda.Event()
            ev.record(self.s_front)
        with torch.cuda.stream(self.s_back):
            self.s_back.wait_event(ev)
            lat = back_half(adapter, feats, steps_unet, seed)
        self.keep.append((feats, lat))          # keep tensors alive until both streams are drained

    def drain(self):
        self.s_front.synchronize()
        self.s_back.synchronize()
        out = [l for _, l in self.keep]
        self.keep = []
        return out


def gemm_roofline(vit, agent, adapter, tok, inp, steps_unet, n_text):
    """Instrumented (eager, un-graphed) generation with HIP events around EVERY sx_gemm launch on the launch stream:
    per-launch algorithmic FLOPs (2·M·N·K, conv: 2·M·N·9·Cin) and duration. The GEMM/implicit-conv kernel is the
    dominant kernel (≈85 % of algorithmic FLOPs)."""
    from seedx_amd import _lib, ops
    lib = _lib.load()
    real = lib.sx_gemm
    rec = []

    class Hook:
        def __call__(self, args_ref, stream):
            a = args_ref._obj
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = real(args_ref, stream)
            e.record()
            n_out = a.N // 2 if a.glu else a.N
            n_st = a.n_valid if a.n_valid else n_out
            a_bytes = 2.0 * (a.B * a.Hin * a.Win * a.Cin if a.a_mode == 1 else a.M * a.K)   # operands once + output once
            byt = a_bytes + 2.0 * a.N * a.K + a.M * n_st * (4.0 if a.out_dtype == 2 else 2.0) \
                + (4.0 * a.M * n_st if a.residual else 0.0)
            rec.append((2.0 * a.M * a.N * a.K, s, e, byt))
            return r

    agent.use_graph, adapter._loop.use_graph = False, False
    lib.sx_gemm = Hook()
    try:
        one_generation(vit, agent, adapter, tok, inp, steps_unet, n_text, seed=1)
        torch.cuda.synchronize()
    finally:
        lib.sx_gemm = real
        agent.use_graph, adapter._loop.use_graph = True, True
    ms = [s.elapsed_time(e) for _, s, e, _ in rec]
    fl = sum(r_[0] for r_ in rec)
    alg_bytes = sum(r_[3] for r_ in rec)
    tot_s = sum(ms) * 1e-3
    n = len(rec)
    # traffic: bytes per launch from the rocprofv3 --pmc passes of THIS command (tools/bench_pmc_traffic.py; graph replay
    # crashes the counter collection on this pool, so the passes run the same step with eager launches). Only quoted
    # when the stored profile was taken at the batch size being run.
    traffic, tnote = None, "no PMC profile for this batch size"
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", "r1_bench_pmc_traffic.json")))
        if prof.get("batch_per_gpu") == BATCH:
            traffic, tnote = prof["traffic_bytes_per_launch"], prof["method"]
    except (OSError, ValueError, KeyError):
        pass
    return {"bound": "mfma", "achieved": fl / tot_s / 1e12, "peak": PEAK_TFLOPS_16BIT, "unit": "TFLOP/s",
            "frac": fl / tot_s / 1e12 / PEAK_TFLOPS_16BIT, "traffic": traffic, "traffic_unit": "bytes/launch",
            "traffic_source": tnote, "algorithmic_bytes_per_launch": alg_bytes / n,
            "kernel": "sxk_gemm::gemm_kernel<*>",
            "launches_per_step": n, "avg_launch_us": tot_s / n * 1e6, "avg_launch_gflop": fl / n / 1e9,
            "gemm_time_s_per_step": tot_s}


def cpu_baseline():
    """Bounded sample of the CPU oracle ("port": oracle/restated*.py, fp32 torch on the host cores), extrapolated by
    layer count / algorithmic FLOPs to one full generation."""
    from oracle import restated, restated_unet as ru, weights
    torch.set_num_threads(os.cpu_count())
    cores = torch.get_num_threads()
    g = torch.Generator().manual_seed(0)
    t_total = 0.0
    # (1) one of 48 ViT blocks at full width, B = 2 crops
    W, heads, L = 1664, 16, 1024
    import torch.nn.functional as F
    x = torch.randn(2, L, W, generator=g)
    wq, wo = torch.randn(3 * W, W, generator=g) * 0.02, torch.randn(W, W, generator=g) * 0.02
    w1, w2 = torch.randn(8192, W, generator=g) * 0.02, torch.randn(W, 8192, generator=g) * 0.02

    def vit_block(x):
        h = F.layer_norm(x, (W,))
        q, k, v = F.linear(h, wq).view(2, L, heads, 3 * 104).split(104, dim=-1)
        a = torch.softmax((q.permute(0, 2, 1, 3) / math.sqrt(104)) @ k.permute(0, 2, 3, 1), -1) @ v.permute(0, 2, 1, 3)
        x = x + F.linear(a.permute(0, 2, 1, 3).reshape(2, L, W), wo)
        return x + F.linear(F.gelu(F.linear(F.layer_norm(x, (W,)), w1)), w2)
    vit_block(x)
    t0 = time.time(); vit_block(x); t_vit_layer = time.time() - t0
    t_total += t_vit_layer * 48 * (FLOP_VIT_CROP * 2 / (2 * 2 * 42.8e9 * 48))   # + attn_pool/proj share by FLOPs
    # (2) one of 40 Llama layers: prefill T=165 and 4 cached decode steps
    cfg = dict(weights.FULL_LLM, num_hidden_layers=1, vocab_size=512)
    sd = weights.llama_sd(cfg)
    xe = torch.randn(1, 165, 5120, generator=g)
    t0 = time.time(); _, past, _ = restated.llama_forward(sd, cfg, xe); t_pre = time.time() - t0
    t0 = time.time()
    for _ in range(4):
        _, past, _ = restated.llama_forward(sd, cfg, xe[:, :1], past)
    t_dec = (time.time() - t0) / 4
    n_new = 128
    t_total += 40 * (t_pre + n_new * t_dec)
    # (3) SDXL UNet: mid-block resnet + one of its 10 transformer layers at 32x32, CFG batch 2 → scale by FLOPs
    ucfg = ru.FULL_UNET
    shapes = ru.unet_param_shapes(ucfg)
    usd = {}
    for k, s in shapes.items():
        if k.startswith("mid_block.resnets.0") or k.startswith("mid_block.attentions.0.norm") or \
                k.startswith("mid_block.attentions.0.proj") or k.startswith("mid_block.attentions.0.transformer_blocks.0."):
            usd[k] = torch.randn(s, generator=g) * (0.02 if len(s) > 1 else 1.0)
    xs = torch.randn(2, 1280, 32, 32, generator=g)
    emb = torch.randn(2, 1280, generator=g)
    ehs = torch.randn(2, 64, 2048, generator=g)
    t0 = time.time()
    h = ru._resnet(usd, "mid_block.resnets.0", xs, emb, 32)
    ru._transformer(usd, "mid_block.attentions.0", h, ehs, 20, 1, 32)
    t_u = time.time() - t0
    C, HW, B = 1280, 1024, 2
    fl_sample = B * (2 * 2 * HW * C * 9 * C + 2 * HW * C * C * 2            # resnet convs + proj_in/out
                     + 2 * HW * C * (3 * C + C + C + C + 8 * C + 4 * C)        # qkv, out, q2, out2, geglu, ff2
                     + 4 * HW * HW * C + 4 * HW * 64 * C + 2 * 2 * 64 * 2048 * C)
    t_total += 50 * t_u * (2 * FLOP_UNET_SAMPLE / fl_sample)
    return {"value": 1.0 / t_total, "unit": "gens/s", "cores": cores, "kind": "port",
            "sample": "oracle fp32 torch on host cores: 1 of 48 ViT-G blocks (B=2) %.2fs, 1 of 40 Llama-13B-dim layers "
                      "(prefill T=165 %.2fs, cached decode %.3fs/token), SDXL mid-block resnet + 1 transformer layer @32x32 "
                      "CFG-2 %.2fs; extrapolated by layer count / algorithmic FLOPs to one generation (%.0f s)"
                      % (t_vit_layer, t_pre, t_dec, t_u, t_total)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--unet-steps", type=int, default=50)
    ap.add_argument("--text-tokens", type=int, default=61)
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of HIP-graph replay (profiling aid)")
    ap.add_argument("--overlap", type=int, default=0,
                    help="1: pipeline consecutive requests on two streams (LLM decode of request i+1 under the UNet of request i)")
    ap.add_argument("--batch", type=int, default=16,
                    help="independent generations processed together per GPU per step (1 = single-request latency mode)")
    ap.add_argument("--no-vae", action="store_true", help="stop at the denoised latents (no VAE decode)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    a = ap.parse_args()
    global BATCH
    BATCH = a.batch
    global USE_VAE
    USE_VAE = not a.no_vae
    from seedx_amd import dist_utils as du
    ctx = du.init("nccl")                       # RCCL over xGMI; only barrier + max-reduce of the wall time
    rank, world, local = ctx.rank, ctx.world, ctx.local
    assert world == a.gpus or world == 1, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    tok = BenchTokenizer()
    with torch.no_grad():
        vit, agent, adapter = build_models(dev, dtype)
        inp = make_inputs(dev)
        if a.no_graph:
            agent.use_graph = False
            adapter._loop.use_graph = False
        seeds = du.shard_seeds(ctx, a.steps)      # independent requests, round-robin over ranks
        pipe = Pipeline(vit, agent, adapter, tok, inp, a.unet_steps, a.text_tokens) if a.overlap else None

        def run(seed_list):
            if pipe is None:
                for sd_ in seed_list:
                    one_generation(vit, agent, adapter, tok, inp, a.unet_steps, a.text_tokens, seed=sd_)
            else:
                for sd_ in seed_list:
                    pipe.submit(sd_)
                pipe.drain()
        run([100 + i for i in range(a.warmup)])
        torch.cuda.synchronize()
        du.barrier(ctx)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(seeds)
        torch.cuda.synchronize()
        du.barrier(ctx)
        torch.cuda.synchronize()
        dt = du.max_over_ranks(ctx, time.perf_counter() - t0)
        roof = None
        if rank == 0 and not a.no_roofline:
            roof = gemm_roofline(vit, agent, adapter, tok, inp, a.unet_steps, a.text_tokens)
    if rank == 0:
        total = du.total_units(ctx, a.steps) * a.batch
        rec = {"metric": "end-to-end generations/sec (img-in -> txt + 1024px-img-out)", "value": total / dt,
               "unit": "gens/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": a.dtype, "data": "synthetic (seeded random image/prompt, random-init weights of the real dims)",
               "config": {"workload": "SEED-X-I: 1x448px image (2 crops ViT-G) -> 165-token prefill -> 128 greedy tokens "
                                      "(%d text + <img> + 64 forced + </img>) -> %d-step SDXL-UNet CFG-2 de-tokenize "
                                      "@1024x1024 -> %s" % (a.text_tokens, a.unet_steps,
                                                            "SDXL VAE decode to a [3,1024,1024] image" if USE_VAE
                                                            else "latents (VAE decode skipped)"),
                          "parallelism": "replica x%d (independent generations, no data-path collective)" % world,
                          "batch_per_gpu": a.batch,
                          "request_pipelining": bool(a.overlap)},
               "flops_per_generation": 2 * FLOP_VIT_CROP + (165 + 128) * FLOP_LLM_TOKEN + 2 * a.unet_steps * FLOP_UNET_SAMPLE
               + (FLOP_VAE_DECODE if USE_VAE else 0.0),
               "generations_per_step": a.batch}
        rec["model_tflops"] = rec["flops_per_generation"] * rec["value"] / world / 1e12
        if roof is not None:
            rec["roofline"] = roof
        if not a.no_cpu_baseline and world == 1:
            try:
                rec["cpu_baseline"] = cpu_baseline()
            except Exception as ex:  # the oracle is optional infrastructure; never fail the measurement on it
                rec["cpu_baseline"] = {"error": repr(ex)}
        print(json.dumps(rec), flush=True)
    du.barrier(ctx)                 # rank 0 ran the instrumented roofline pass alone: tear the group down together
    if world > 1:
        torch.cuda.synchronize()
    du.finalize(ctx)


if __name__ == "__main__":
    main()
