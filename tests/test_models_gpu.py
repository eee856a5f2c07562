"""Module-level parity on the GPU: the HIP-backed modules vs the CPU oracle (oracle/restated.py, itself pinned
against the reference's own modules) with identical seeded weights and inputs.

Metric: relative L2 error ‖x − x_ref‖₂ / ‖x_ref‖₂ against the fp32 oracle (SURVEY.md §7 hard part 5). The GEMM
operands are rounded to 16 bit (that is the reference's own inference precision: fp16, eval_*.py), so the floor is
set by operand rounding: tolerances are 2e-3 for fp16 and 1.6e-2 for bf16 (8x coarser mantissa) per module."""
import math

import pytest
import torch

from oracle import restated, weights

pytestmark = pytest.mark.gpu
TOL = {torch.float16: 2e-3, torch.bfloat16: 1.6e-2}


def relerr(x, ref):
    x, ref = x.float().cpu(), ref.float().cpu()
    return ((x - ref).norm() / ref.norm()).item()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("cfg", [weights.MINI_VIT, weights.MINI_VIT_104], ids=["hd128", "hd104"])
def test_vit_vs_oracle(dev, dtype, cfg):
    from seedx_amd.visual_encoder import VisionTransformerWithAttnPool
    sd = weights.vit_sd(cfg)
    x = torch.randn(2, 3, cfg["image_size"], cfg["image_size"], generator=torch.Generator().manual_seed(0))
    ref = restated.vit_forward(sd, cfg, x)
    m = VisionTransformerWithAttnPool(**cfg)
    m.load_state_dict(sd)
    m.eval().to(dev, dtype=dtype)
    y = m(x)
    assert y.shape == ref.shape and y.dtype == dtype
    e = relerr(y, ref)
    print(f"vit {dtype} relerr {e:.3e}")
    assert e < TOL[dtype]


def test_vit_missing_key_is_loud(dev):
    from seedx_amd.visual_encoder import VisionTransformerWithAttnPool
    cfg = weights.MINI_VIT
    sd = weights.vit_sd(cfg)
    del sd["ln_post.bias"]
    with pytest.raises(KeyError):
        VisionTransformerWithAttnPool(**cfg).load_state_dict(sd)


# ---------------------------------------------------------------------------------------------------------
# Path B — LLM prefill / decode / ContinuousLVLM.generate
# ---------------------------------------------------------------------------------------------------------
class StubTokenizer:
    """Minimal tokenizer with the reference's special tokens: <img>=400, <img_00000..63>=401..464, </img>=465."""
    eos_token_id = 2

    def encode(self, s, add_special_tokens=False):
        import re
        out = []
        for tok in re.findall(r"<img_\d{5}>|<img>|</img>|\S+", s):
            if tok == "<img>":
                out.append(400)
            elif tok == "</img>":
                out.append(465)
            elif tok.startswith("<img_"):
                out.append(401 + int(tok[5:10]))
            else:
                out.append(int(tok))
        return out

    def decode(self, ids, skip_special_tokens=False):
        return " ".join(str(int(i)) for i in ids)


def _build_lvlm(dev, dtype, sd_llm, sd_agent, cfg, vit_dim):
    from seedx_amd.llama import LlamaForCausalLM
    from seedx_amd.seed_x import ContinuousLVLM
    from seedx_amd.visual_encoder import Resampler
    llm = LlamaForCausalLM(dict(cfg), max_cache_len=512)
    llm.load_state_dict(sd_llm)
    H = cfg["hidden_size"]
    agent = ContinuousLVLM(llm, Resampler(4, H, 2, kv_dim=vit_dim), Resampler(4, vit_dim, 2, kv_dim=H), add_patch_pos=True)
    agent.load_state_dict(sd_agent)
    agent.eval().to(dev, dtype=dtype)
    return agent


def _lvlm_inputs(cfg, vit_dim):
    g = torch.Generator().manual_seed(5)
    # prompt: BOS, 3 text, <patch-ish> 16 image slots (crop 0), 16 image slots (crop 1), 5 text tokens
    ids = [1, 11, 12, 13] + [0] * 16 + [0] * 16 + [21, 22, 23, 24, 25]
    mask = torch.zeros(1, len(ids), dtype=torch.bool)
    mask[0, 4:36] = True
    image_embeds = torch.randn(2, 36, vit_dim, generator=g)          # 2 crops x 36 ViT tokens (6x6 → 4x4 queries)
    emask = torch.tensor([True, True])
    ppos = torch.tensor([[0.0, 0.0], [0.5, 0.5]])
    return ids, mask, image_embeds, emask, ppos


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_llm_generate_vs_oracle(dev, dtype):
    cfg = weights.MINI_LLM
    vit_dim = 128
    sd_llm = weights.llama_sd(cfg)
    sd_agent = weights.agent_sd(cfg, vit_dim, in_grid=4, out_grid=4)
    ids, mask, image_embeds, emask, ppos = _lvlm_inputs(cfg, vit_dim)
    img_ids = list(range(400, 466))
    rc = {"in_heads": 2, "out_heads": 2}
    # make <img> the oracle's first greedy choice: align lm_head[<img>] with the last prompt position's final state
    pre = restated.lvlm_generate(sd_llm, sd_agent, cfg, rc, ids, image_embeds, emask, mask, ppos, img_ids, 400, 465, 0,
                                 16, return_prefill=True)
    hn_last = pre["hidden"][0, -1]
    sd_llm["lm_head.weight"][400] = 4.0 * hn_last / hn_last.norm()
    agent = _build_lvlm(dev, dtype, sd_llm, sd_agent, cfg, vit_dim)
    tok = StubTokenizer()
    max_new = 30
    for chunked in (True, False):
        agent.chunk_forced_image_tokens = chunked
        out = agent.generate(tok, input_ids=[ids], image_embeds=image_embeds.to(dev), embeds_cmp_mask=emask,
                             ids_cmp_mask=mask, patch_positions=ppos, max_new_tokens=max_new, num_img_gen_tokens=16,
                             eos_token_id=None)
        new = out["generate_ids"].tolist()
        img16 = [400] + list(range(401, 417)) + [465]
        assert len(new) == max_new and new[:18] == img16, new
        # oracle under teacher forcing with OUR ids: every step must agree with the oracle's argmax unless the oracle's
        # own top-2 gap is within 16-bit noise of its score scale
        trace = []
        ref = restated.lvlm_generate(sd_llm, sd_agent, cfg, rc, ids, image_embeds, emask, mask, ppos,
                                     [400] + list(range(401, 417)) + [465], 400, 465, max_new, 16, None, dtype, new, trace)
        tol_gap = 0.02 if dtype == torch.float16 else 0.12
        for step, (o_arg, forced, gap, std) in enumerate(trace):
            assert o_arg == forced or gap < tol_gap * max(std, 1.0), (step, o_arg, forced, gap, std)
        e_h = relerr(out["last_hidden_states"], ref["last_hidden"])
        e_f = relerr(out["img_gen_feat"], ref["img_gen_feat"])
        print(f"llm {dtype} chunked={chunked} hidden relerr {e_h:.3e} img_feat relerr {e_f:.3e}")
        assert out["has_img_output"] and out["num_gen_imgs"] == 1 and out["img_gen_feat"].shape == (1, 16, vit_dim)
        assert e_h < 2 * TOL[dtype] and e_f < 2 * TOL[dtype]
        # text = generated ids minus <img> and minus the 16 ids before each </img> (seed_x.py:199-216; </img> itself stays)
        keep = [True] * len(new)
        for e in [i for i, t in enumerate(new) if t == 465]:
            for j in range(e - 16, e):
                keep[j] = False
        keep = [k and t != 400 for k, t in zip(keep, new)]
        assert out["text"] == " ".join(str(t) for t, k in zip(new, keep) if k)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_llm_prefill_logits_vs_oracle(dev, dtype):
    from seedx_amd.llama import LlamaForCausalLM
    cfg = weights.MINI_LLM
    sd = weights.llama_sd(cfg)
    x = torch.randn(1, 37, cfg["hidden_size"], generator=torch.Generator().manual_seed(6)) * 0.5
    logits_ref, _, hn_ref = restated.llama_forward(sd, cfg, x, table_dtype=dtype)
    llm = LlamaForCausalLM(dict(cfg), max_cache_len=128)
    llm.load_state_dict(sd)
    llm.eval().to(dev, dtype=dtype)
    out = llm(inputs_embeds=x.to(dev), output_hidden_states=True)
    e_l = relerr(out["logits"][0, -1], logits_ref[0, -1])
    e_h = relerr(out["hidden_states"][-1], hn_ref)
    print(f"llm prefill {dtype} logits relerr {e_l:.3e} hidden relerr {e_h:.3e}")
    assert e_l < TOL[dtype] and e_h < TOL[dtype]
    # graph-replayed single-token steps == eager single-token steps (same kernels → bit identical)
    img_ids = torch.arange(400, 466, dtype=torch.int32, device=dev)
    res = []
    for use_graph in (False, True):
        llm(inputs_embeds=x.to(dev))
        llm._P["cur"].fill_(7)
        llm._P["step"].fill_(1)
        out_ids = torch.full((1, 16), -1, dtype=torch.int32, device=dev)
        hid = torch.zeros((1, 16, cfg["hidden_size"]), device=dev)
        for _ in range(6):
            llm.decode_step(img_ids, out_ids, hid, use_graph=use_graph)
        res.append((out_ids.clone(), hid.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert (res[0][0][0, 1:7] >= 0).all()


# ---------------------------------------------------------------------------------------------------------
# Path C — ResamplerXLV2, UNet, denoise loops
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("n_tokens", [16, 36])
def test_resampler_xlv2_vs_oracle(dev, dtype, n_tokens):
    from seedx_amd.detokenizer import ResamplerXLV2
    cfg = weights.MINI_XLV2
    sd = weights.xlv2_sd(cfg, pre="resampler.")
    x = torch.randn(2, n_tokens, cfg["embedding_dim"], generator=torch.Generator().manual_seed(8))
    pe_ref, pool_ref = restated.resampler_xlv2_forward(sd, cfg, x)
    m = ResamplerXLV2(normalize=False, **cfg)
    m.load_state_dict(sd, prefix="resampler.")
    m.to(dev, dtype)
    pe, pool = m(x.to(dev))
    e1, e2 = relerr(pe, pe_ref), relerr(pool, pool_ref)
    print(f"xlv2 {dtype} prompt relerr {e1:.3e} pooled relerr {e2:.3e}")
    assert e1 < TOL[dtype] and e2 < TOL[dtype]


def _unet_inputs(cfg, B, hw=16, seed=9):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, cfg["in_channels"], hw, hw, generator=g)
    ehs = torch.randn(B, 16, cfg["cross_attention_dim"], generator=g)
    te = torch.randn(B, cfg["pooled_dim"], generator=g)
    tid = torch.tensor([[1024.0, 1024, 0, 0, 1024, 1024]] * B)
    return x, ehs, te, tid


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("in_ch,B", [(4, 2), (8, 3)])
def test_unet_forward_vs_oracle(dev, dtype, in_ch, B):
    from oracle import restated_unet as ru
    from seedx_amd.unet import UNet2DConditionModel
    cfg = dict(ru.MINI_UNET, in_channels=in_ch)
    sd = ru.unet_sd(cfg)
    x, ehs, te, tid = _unet_inputs(cfg, B)
    ref = ru.unet_forward(sd, cfg, x, 981.0, ehs, te, tid)
    m = UNet2DConditionModel(**cfg)
    m.load_state_dict(sd)
    m.to(dev, dtype)
    out = m(x.to(dev), 981.0, ehs.to(dev), added_cond_kwargs={"text_embeds": te.to(dev), "time_ids": tid.to(dev)},
            return_dict=False)[0]
    e = relerr(out, ref)
    print(f"unet {dtype} in_ch={in_ch} relerr {e:.3e}  (ref std {ref.std():.3f})")
    assert out.shape == ref.shape and e < 2 * TOL[dtype]
    assert m(x.to(dev), 981.0, ehs.to(dev), added_cond_kwargs={"text_embeds": te.to(dev), "time_ids": tid.to(dev)}).sample.shape == ref.shape


@pytest.mark.parametrize("mode", [0, 1])
def test_denoise_loop_vs_oracle(dev, mode):
    """5-step CFG + Euler loop (graph replayed) against the oracle loop driving the oracle UNet."""
    from oracle import restated_unet as ru
    from seedx_amd.detokenizer import EulerDiscreteScheduler, _DenoiseLoop
    from seedx_amd.unet import UNet2DConditionModel
    dtype = torch.float16
    cfg = dict(ru.MINI_UNET, in_channels=4 if mode == 0 else 8)
    sd = ru.unet_sd(cfg)
    g = torch.Generator().manual_seed(10)
    pe, ne = torch.randn(1, 16, 128, generator=g), torch.randn(1, 16, 128, generator=g)
    pp, npool = torch.randn(1, 128, generator=g), torch.randn(1, 128, generator=g)
    tid = torch.tensor([[1024.0, 1024, 0, 0, 1024, 1024]])
    steps = 5
    _, _, init = ru.euler_tables(steps)
    lat0 = torch.randn(1, 4, 16, 16, generator=g) * init
    il = torch.randn(1, 4, 16, 16, generator=g)
    fn = lambda s, t, e, p, ti: ru.unet_forward(sd, cfg, s, t, e, p, ti)
    m = UNet2DConditionModel(**cfg)
    m.load_state_dict(sd)
    m.to(dev, dtype)
    sch = EulerDiscreteScheduler()
    for use_graph in (False, True):
        loop = _DenoiseLoop(m, use_graph=use_graph)
        if mode == 0:
            ref = ru.t2i_loop(fn, lat0, pe, ne, pp, npool, tid, steps)
            out = loop.run(0, lat0, torch.cat([ne, pe]), torch.cat([npool, pp]), tid.repeat(2, 1), sch, steps, 7.5)
        else:
            il3 = torch.cat([il, il, torch.zeros_like(il)])
            ref = ru.edit_loop(fn, lat0, il3, pe, ne, pp, npool, tid, steps)
            out = loop.run(1, lat0, torch.cat([pe, ne, ne]), torch.cat([pp, npool, npool]), tid.repeat(3, 1), sch, steps,
                           7.5, 1.5, il3)
        e = relerr(out, ref)
        print(f"denoise mode={mode} graph={use_graph} relerr {e:.3e}")
        assert e < 5e-3
        if use_graph:   # second run through the cached graph with different conditioning must track the oracle too
            pe2 = pe * 0.5
            if mode == 0:
                ref2 = ru.t2i_loop(fn, lat0, pe2, ne, pp, npool, tid, steps)
                out2 = loop.run(0, lat0, torch.cat([ne, pe2]), torch.cat([npool, pp]), tid.repeat(2, 1), sch, steps, 7.5)
            else:
                ref2 = ru.edit_loop(fn, lat0, il3, pe2, ne, pp, npool, tid, steps)
                out2 = loop.run(1, lat0, torch.cat([pe2, ne, ne]), torch.cat([pp, npool, npool]), tid.repeat(3, 1), sch,
                                steps, 7.5, 1.5, il3)
            assert relerr(out2, ref2) < 5e-3
