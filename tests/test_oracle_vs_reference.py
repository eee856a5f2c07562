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
