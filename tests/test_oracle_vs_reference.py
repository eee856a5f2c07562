"""Pins the CPU oracle (oracle/restated.py) against the reference's OWN modules executed on CPU.

Runs only where /root/reference exists (the build container); on the GPU box the committed golden fixtures in
tests/golden/ (generated from the same reference modules by oracle/gen_golden.py) take over."""
import pytest
import torch

from oracle import refshim, restated, weights

pytestmark = pytest.mark.skipif(not refshim.available(), reason="/root/reference not present")


def _rel(a, b):
    return ((a - b).norm() / b.norm()).item()


@pytest.mark.parametrize("cfg", [weights.MINI_VIT, weights.MINI_VIT_104])
def test_vit_matches_reference(cfg):
    ref = refshim.reference_modules()["VisionTransformerWithAttnPool"](**cfg).eval()
    sd = weights.vit_sd(cfg)
    missing, unexpected = ref.load_state_dict(sd, strict=True)
    x = torch.randn(2, 3, cfg["image_size"], cfg["image_size"], generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        y_ref = ref(x)
    y = restated.vit_forward(sd, cfg, x)
    assert y.shape == y_ref.shape and _rel(y, y_ref) < 2e-5


def _ref_llama(cfg, sd):
    from transformers import LlamaConfig
    m = refshim.reference_modules()["LlamaForCausalLM"](LlamaConfig(**cfg)).eval()
    full = dict(m.state_dict())
    full.update(sd)
    m.load_state_dict(full, strict=True)
    return m


def test_llama_prefill_and_decode_match_reference():
    cfg = weights.MINI_LLM
    sd = weights.llama_sd(cfg)
    m = _ref_llama(cfg, sd)
    x = torch.randn(1, 11, cfg["hidden_size"], generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        o = m(inputs_embeds=x, attention_mask=torch.ones(1, 11, dtype=torch.long), use_cache=True,
              output_hidden_states=True, return_dict=True)
    logits, past, hn = restated.llama_forward(sd, cfg, x)
    assert _rel(logits, o.logits) < 2e-5 and _rel(hn, o.hidden_states[-1]) < 2e-5
    assert _rel(past[1][0], o.past_key_values[1][0]) < 2e-5
    # one cached decode step (q_len == 1, unmasked attention branch of the reference)
    tok = torch.tensor([[17]])
    with torch.no_grad():
        o2 = m(input_ids=tok, attention_mask=torch.ones(1, 12, dtype=torch.long), past_key_values=o.past_key_values,
               use_cache=True, output_hidden_states=True, return_dict=True)
    l2, _, h2 = restated.llama_forward(sd, cfg, sd["model.embed_tokens.weight"][tok], past)
    assert _rel(l2, o2.logits) < 2e-5 and _rel(h2, o2.hidden_states[-1]) < 2e-5


def test_llama_ckpt16_fixture_reproduces_from_the_live_reference():
    """tests/golden/llama_mini_ckpt16.npz is what the reference's own module computes (gen_golden.run_reference_llama_ckpt16 re-run here)."""
    import os
    import numpy as np
    from oracle import gen_golden as gg
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "llama_mini_ckpt16.npz"))
    live = gg.run_reference_llama_ckpt16()["llama_mini_ckpt16.npz"]
    for k in gold.files:
        a = live[k].detach().numpy() if torch.is_tensor(live[k]) else np.asarray(live[k])
        assert a.shape == gold[k].shape and np.allclose(a, gold[k], rtol=1e-5, atol=1e-6), k
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "modules_mini_ckpt16.npz"))
    live = gg.run_reference_modules_ckpt16()["modules_mini_ckpt16.npz"]
    for k in gold.files:
        a = live[k].detach().numpy() if torch.is_tensor(live[k]) else np.asarray(live[k])
        assert a.shape == gold[k].shape and np.allclose(a, gold[k], rtol=1e-5, atol=1e-6), k


def test_logits_rule_matches_reference_processor():
    """generation.py:19-31 with a fake tokenizer whose image-token ids are 400..465."""
    cls = refshim.reference_modules()["AutoImageTokenGenerationProcessor"]
    ids = list(range(400, 466))

    class Tok:
        def encode(self, s, add_special_tokens=False):
            return ids
    proc = cls(Tok(), num_img_gen_tokens=64)
    g = torch.Generator().manual_seed(2)
    for last in (5, 400, 433, 464, 465):
        scores = torch.randn(1, 500, generator=g) - 3.0
        ref = proc(torch.tensor([[1, 2, last]]), scores.clone())
        mine = restated.logits_rule(last, scores[0].clone(), ids)
        assert torch.equal(ref[0], mine)
        assert int(ref[0].argmax()) == int(mine.argmax())


def test_resampler_and_xlv2_match_reference():
    mods = refshim.reference_modules()
    g = torch.Generator().manual_seed(3)
    # LLM-side Resampler (agent_seed_x_i.yaml geometry, mini dims; eps = nn.LayerNorm default 1e-5)
    r = mods["Resampler"](grid_size=4, embed_dim=320, num_heads=2, kv_dim=256).eval()
    sd = weights.resampler_sd(weights._g(7), "", 4, 320, 256)
    r.load_state_dict(sd, strict=True)
    x = torch.randn(2, 16, 256, generator=g)
    with torch.no_grad():
        y_ref = r(x)
    assert _rel(restated.resampler_forward(sd, "", x, 2, 1e-5), y_ref) < 2e-5
    cfg = weights.MINI_XLV2
    m = mods["ResamplerXLV2"](normalize=False, **cfg).eval()
    sdx = weights.xlv2_sd(cfg, pre="")
    m.load_state_dict(sdx, strict=True)
    x = torch.randn(2, 24, cfg["embedding_dim"], generator=g)
    with torch.no_grad():
        pe, pooled = m(x)
    pe2, pooled2 = restated.resampler_xlv2_forward(sdx, cfg, x, pre="")
    assert _rel(pe2, pe) < 2e-5 and _rel(pooled2, pooled) < 2e-5


# ---- path C: the reference's OWN adapters + edit pipeline (executed over oracle/diffusers_shim.py) -------------------
def test_detok_goldens_are_reproducible_and_pin_the_restated_adapter():
    """(1) re-running the reference adapters reproduces the committed tests/golden/{t2i,edit}_mini.npz bit for bit;
    (2) oracle/restated_adapter.py + restated_unet.{t2i,edit}_loop (the restatements every GPU parity test uses) agree
    with what the reference code computed: get_image_embeds branches, 5-step trajectories, VAE-encoded source."""
    import os
    import numpy as np
    from oracle import gen_golden as gg, restated_adapter as ra, restated_unet as ru, restated_vae as rv
    live = gg.run_reference_detok()
    gold_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for name, arrs in live.items():
        gold = np.load(os.path.join(gold_dir, name))
        for k, v in arrs.items():
            assert np.array_equal(gold[k], v.numpy()), f"{name}:{k} differs from the committed fixture"
    t2i, edit = live["t2i_mini.npz"], live["edit_mini.npz"]
    sd_vit, sd_x = weights.vit_sd(gg.DETOK_VIT), weights.xlv2_sd(gg.DETOK_XLV2)
    out = ra.get_image_embeds(sd_vit, gg.DETOK_VIT, sd_x, gg.DETOK_XLV2, image_tensor=t2i["image_tensor"])
    for o, k in zip(out, ("tensor_prompt", "tensor_prompt_neg", "tensor_pooled", "tensor_pooled_neg")):
        assert _rel(o, t2i[k]) < 2e-5, k
    out = ra.get_image_embeds(sd_vit, gg.DETOK_VIT, sd_x, gg.DETOK_XLV2, image_embeds=t2i["feats"], vit_down=True)
    for o, k in zip(out, ("embeds_prompt", "embeds_prompt_neg", "embeds_pooled", "embeds_pooled_neg")):
        assert _rel(o, t2i[k]) < 2e-5, k
    u4, u8 = gg.detok_unet_cfg(4), gg.detok_unet_cfg(8)
    sd4, sd8 = ru.unet_sd(u4), ru.unet_sd(u8)
    kw = dict(height=128, width=128)
    lat = ra.adapter_generate(sd_vit, gg.DETOK_VIT, sd_x, gg.DETOK_XLV2, sd4, u4, t2i["noise"], 5,
                              image_embeds=t2i["feats"], **kw)
    assert _rel(lat, t2i["latents"]) < 5e-5
    assert torch.equal(t2i["latents_traj"][-1], t2i["latents"])
    lat = ra.adapter_generate(sd_vit, gg.DETOK_VIT, sd_x, gg.DETOK_XLV2, sd8, u8, edit["noise"], 5,
                              image_embeds=edit["feats"], image_latents=edit["image_latents"], **kw)
    assert _rel(lat, edit["latents"]) < 5e-5
    lat = ra.adapter_generate(sd_vit, gg.DETOK_VIT, sd_x, gg.DETOK_XLV2, sd8, u8, edit["noise"], 3,
                              image_embeds=edit["feats"], image_latents=torch.zeros(1, 4, 16, 16), **kw)
    assert _rel(lat, edit["latents_no_image"]) < 5e-5
    il = ra.edit_image_latents(rv.vae_encoder_sd(gg.DETOK_VAE), gg.DETOK_VAE, edit["src_image"])
    lat = ra.adapter_generate(sd_vit, gg.DETOK_VIT, sd_x, gg.DETOK_XLV2, sd8, u8, edit["noise"], 2,
                              image_embeds=edit["feats"], image_latents=il, **kw)
    assert _rel(lat, edit["latents_from_rgb"]) < 5e-5
    lat2 = ra.adapter_generate(sd_vit, gg.DETOK_VIT, sd_x, gg.DETOK_XLV2, sd8, u8, edit["noise"], 2,
                               image_embeds=edit["feats"], image_latents=edit["image_latents"], **kw)
    img = ra.decode_to_pt(rv.vae_sd(gg.DETOK_VAE), gg.DETOK_VAE, lat2)
    assert (img - edit["image_pt"]).abs().max() < 1e-4


# ---- path B orchestration: the reference's OWN ContinuousLVLM.generate over the HF-4.30.2 greedy stand-in -------------------
def test_lvlm_generate_golden_is_reproducible():
    """Re-executes seed_x.py:130-223 (+ modeling_llama_xformer.py:748-779, generation.py:19-31) through
    oracle/hf_generate_shim.py and gets the committed tests/golden/lvlm_generate_mini.npz back bit for bit."""
    import os
    import numpy as np
    from oracle import gen_golden as gg
    live = gg.run_reference_generate()["lvlm_generate_mini.npz"]
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lvlm_generate_mini.npz"))
    assert set(live) == set(gold.files)
    for k, v in live.items():
        v = v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)
        assert np.array_equal(gold[k], v), f"{k} differs from the committed fixture"


def test_lvlm_generate_ckpt16_golden_is_reproducible():
    """The same for tests/golden/lvlm_generate_mini_ckpt16.npz: the reference's generate() on the weights a 16-bit checkpoint holds, with the
    RoPE tables of its fp16 runs (the fixture the HIP path is held to at north_star's 1e-3)."""
    import os
    import numpy as np
    from oracle import gen_golden as gg
    live = gg.run_reference_generate(ckpt16=True)["lvlm_generate_mini_ckpt16.npz"]
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lvlm_generate_mini_ckpt16.npz"))
    assert set(live) == set(gold.files)
    for k, v in live.items():
        v = v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)
        assert np.array_equal(gold[k], v), f"{k} differs from the committed fixture"


def test_generate_shim_drives_the_reference_hooks():
    """The stand-in must call the MODEL's own prepare_inputs_for_generation every step (step 0: inputs_embeds AND input_ids,
    later: the last id only, position = cumsum(mask) - 1) and hand the growing attention_mask back to it."""
    from oracle import gen_golden as gg, hf_generate_shim as hs
    cfg, sd_llm, sd_agent = gg._lvlm_weights()
    m = hs.build_reference_lvlm(cfg, sd_llm, sd_agent, gg.LVLM_VIT_DIM, gg.LVLM_GRID, gg.LVLM_GRID, gg.LVLM_HEADS)
    calls = []
    inner = m.llm.prepare_inputs_for_generation

    def spy(input_ids, inputs_embeds=None, **kw):          # the stand-in inspects the signature for `inputs_embeds`, as 4.30.2 does
        out = inner(input_ids, inputs_embeds=inputs_embeds, **kw)
        calls.append((tuple(input_ids.shape), "inputs_embeds" in out, tuple(out["input_ids"].shape),
                      out["position_ids"].tolist(), tuple(kw["attention_mask"].shape), kw.get("past_key_values") is not None))
        return out
    m.llm.prepare_inputs_for_generation = spy
    with torch.no_grad():
        m.generate(hs.StubTokenizer(), input_ids=[[1, 7, 8, 9]], max_new_tokens=3, dtype=torch.float32, device="cpu")
    assert calls[0] == ((1, 4), True, (1, 4), [[0, 1, 2, 3]], (1, 4), False)
    assert calls[1] == ((1, 5), False, (1, 1), [[4]], (1, 5), True)
    assert calls[2] == ((1, 6), False, (1, 1), [[5]], (1, 6), True)
