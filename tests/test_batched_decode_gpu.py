"""Lock-step batched decode (generate_batch, G = 3 independent requests with different prompts / images) must reproduce
three separate single-request generate() runs token for token, hidden state for hidden state (same kernels on the
same per-sequence data; only the GEMV row count differs) — including a request that emits an image block mid-way."""
import pytest
import torch

from oracle import restated, weights
from tests.test_models_gpu import StubTokenizer, relerr

pytestmark = pytest.mark.gpu


def _agent(dev, dtype, sd_llm, sd_agent, cfg, vit_dim, G, precise=None):
    from seedx_amd.llama import LlamaForCausalLM
    from seedx_amd.seed_x import ContinuousLVLM
    from seedx_amd.visual_encoder import Resampler
    llm = LlamaForCausalLM(dict(cfg), max_cache_len=512, max_batch=G, precise=precise)
    llm.load_state_dict(dict(sd_llm))
    H = cfg["hidden_size"]
    agent = ContinuousLVLM(llm, Resampler(4, H, 2, kv_dim=vit_dim), Resampler(4, vit_dim, 2, kv_dim=H), add_patch_pos=True)
    agent.load_state_dict(sd_agent)
    agent.eval().to(dev, dtype=dtype)
    return agent


def test_generate_batch_equals_single_runs(dev):
    dtype, cfg, vit_dim = torch.float16, weights.MINI_LLM, 128
    sd_llm = weights.llama_sd(cfg)
    sd_agent = weights.agent_sd(cfg, vit_dim, in_grid=4, out_grid=4)
    g = torch.Generator().manual_seed(21)
    reqs = []
    for r, n_text in enumerate((5, 9, 2)):
        ids = [1, 10 + r] + [0] * 16 + [20 + r + i for i in range(n_text)]
        mask = torch.zeros(1, len(ids), dtype=torch.bool)
        mask[0, 2:18] = True
        reqs.append(dict(input_ids=[ids], image_embeds=torch.randn(1, 36, vit_dim, generator=g).to(dev),
                         embeds_cmp_mask=torch.tensor([True]), ids_cmp_mask=mask, patch_positions=torch.tensor([[0.5, 0.5]])))
    tok = StubTokenizer()
    kw = dict(num_img_gen_tokens=16, max_new_tokens=28, eos_token_id=None, force_image_at=4)
    batch = _agent(dev, dtype, sd_llm, sd_agent, cfg, vit_dim, 3).generate_batch(tok, reqs, **kw)
    single = _agent(dev, dtype, sd_llm, sd_agent, cfg, vit_dim, 1)
    for r in range(3):
        one = single.generate_batch(tok, [reqs[r]], **kw)[0]
        assert batch[r]["generate_ids"].tolist() == one["generate_ids"].tolist()
        assert batch[r]["generate_ids"].tolist()[4:22] == [400] + list(range(401, 417)) + [465]
        assert relerr(batch[r]["last_hidden_states"], one["last_hidden_states"]) < 2e-3
        assert relerr(batch[r]["img_gen_feat"], one["img_gen_feat"]) < 2e-3
        assert batch[r]["text"] == one["text"]
    # and the batch agrees with the CPU oracle under teacher forcing (request 1)
    ids = reqs[1]["input_ids"][0]
    trace = []
    ref = restated.lvlm_generate(sd_llm, sd_agent, cfg, {"in_heads": 2, "out_heads": 2}, ids, reqs[1]["image_embeds"].cpu(),
                                 reqs[1]["embeds_cmp_mask"], reqs[1]["ids_cmp_mask"], reqs[1]["patch_positions"],
                                 list(range(400, 466)), 400, 465, 28, 16, None, None, batch[1]["generate_ids"].tolist(), trace)
    assert relerr(batch[1]["last_hidden_states"], ref["last_hidden"]) < 4e-3


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("precise", [False, True])
def test_generate_batch_above_16_sequences(dev, dtype, precise):
    """20 lock-step sequences: the decode step's skinny GEMMs run two 16-row operand blocks per weight fragment (sx_gemv M = 17..32,
    tiled activations [2][K/32][16][32]; precise mode, round 6: FOUR blocks = two planes x two row blocks) — same tokens as
    single-request runs, hidden states within the 16-bit noise in the plain flow (the G >= 5 path is MFMA, the single-request path VALU:
    another summation order) and within fp32 accumulation noise in precise mode."""
    cfg, vit_dim = weights.MINI_LLM, 128
    sd_llm = weights.llama_sd(cfg)
    sd_agent = weights.agent_sd(cfg, vit_dim, in_grid=4, out_grid=4)
    g = torch.Generator().manual_seed(77)
    reqs = []
    for r in range(20):
        ids = [1, 10 + r] + [0] * 16 + [20 + r + i for i in range(2 + r % 5)]
        mask = torch.zeros(1, len(ids), dtype=torch.bool)
        mask[0, 2:18] = True
        reqs.append(dict(input_ids=[ids], image_embeds=torch.randn(1, 36, vit_dim, generator=g).to(dev),
                         embeds_cmp_mask=torch.tensor([True]), ids_cmp_mask=mask, patch_positions=torch.tensor([[0.5, 0.5]])))
    tok = StubTokenizer()
    kw = dict(num_img_gen_tokens=16, max_new_tokens=26, eos_token_id=None, force_image_at=3)
    agent = _agent(dev, dtype, sd_llm, sd_agent, cfg, vit_dim, 20, precise=precise)
    assert agent.llm._pack()["decode_tiled"] and agent.llm.precise == precise
    assert _agent(dev, dtype, sd_llm, sd_agent, cfg, vit_dim, 20).llm.precise      # the default above 16 sequences is the precise mode now
    batch = agent.generate_batch(tok, reqs, **kw)
    single = _agent(dev, dtype, sd_llm, sd_agent, cfg, vit_dim, 1, precise=precise)   # like with like
    tol = (3e-3 if dtype == torch.float16 else 2.4e-2) if not precise else (2e-4 if dtype == torch.float16 else 2e-3)
    same = 0
    for r in (0, 7, 15, 16, 19):                       # rows of both operand blocks
        one = single.generate_batch(tok, [reqs[r]], **kw)[0]
        a, b = batch[r]["generate_ids"].tolist(), one["generate_ids"].tolist()
        assert a[3:21] == [400] + list(range(401, 417)) + [465] and len(a) == len(b) == 26
        n = next((i for i, (u, v) in enumerate(zip(a, b)) if u != v), len(a))   # random-weight logits can be near-ties
        same += n == len(a)
        assert n >= 21, (r, n, a, b)
        assert relerr(batch[r]["last_hidden_states"][:n - 1], one["last_hidden_states"][:n - 1]) < tol
        assert relerr(batch[r]["img_gen_feat"], one["img_gen_feat"]) < tol
    assert same >= 3


def test_uniform_batched_prefill_and_chunk_equal_single_runs(dev):
    """All requests share prompt length and position → the batched RoPE / flash-attention launches (one per layer for the
    whole batch) are taken; the results must still equal separate single-request runs."""
    dtype, cfg, vit_dim = torch.float16, weights.MINI_LLM, 128
    sd_llm = weights.llama_sd(cfg)
    sd_agent = weights.agent_sd(cfg, vit_dim, in_grid=4, out_grid=4)
    g = torch.Generator().manual_seed(33)
    reqs = []
    for r in range(4):
        ids = [1, 10 + r] + [0] * 16 + [20 + r + i for i in range(6)]
        mask = torch.zeros(1, len(ids), dtype=torch.bool)
        mask[0, 2:18] = True
        reqs.append(dict(input_ids=[ids], image_embeds=torch.randn(1, 36, vit_dim, generator=g).to(dev),
                         embeds_cmp_mask=torch.tensor([True]), ids_cmp_mask=mask, patch_positions=torch.tensor([[0.5, 0.5]])))
    tok = StubTokenizer()
    kw = dict(num_img_gen_tokens=16, max_new_tokens=26, eos_token_id=None, force_image_at=3)
    batch = _agent(dev, dtype, sd_llm, sd_agent, cfg, vit_dim, 4).generate_batch(tok, reqs, **kw)
    single = _agent(dev, dtype, sd_llm, sd_agent, cfg, vit_dim, 1)
    for r in range(4):
        one = single.generate_batch(tok, [reqs[r]], **kw)[0]
        assert batch[r]["generate_ids"].tolist() == one["generate_ids"].tolist()
        assert relerr(batch[r]["last_hidden_states"], one["last_hidden_states"]) < 2e-3
        assert relerr(batch[r]["img_gen_feat"], one["img_gen_feat"]) < 2e-3


def test_cross_turn_kv_reuse_equals_full_reprefill(dev):
    """Three-turn conversation (layout of src/data/sft_clm.py:229-276). With reuse_cache the second and third turns
    prefill only their new tokens, and the generated ids / hidden states are identical to re-prefilling the whole
    transcript like the reference does (seed_x.py:184-189). A changed image invalidates the cached prefix."""
    dtype, cfg, vit_dim = torch.float16, weights.MINI_LLM, 128
    sd_llm = weights.llama_sd(cfg)
    sd_agent = weights.agent_sd(cfg, vit_dim, in_grid=4, out_grid=4)
    g = torch.Generator().manual_seed(44)
    img = torch.randn(1, 36, vit_dim, generator=g).to(dev)
    tok = StubTokenizer()
    kw = dict(num_img_gen_tokens=16, max_new_tokens=10, eos_token_id=None)
    reuse = _agent(dev, dtype, sd_llm, sd_agent, cfg, vit_dim, 1)
    full = _agent(dev, dtype, sd_llm, sd_agent, cfg, vit_dim, 1)

    def req(ids):
        mask = torch.zeros(1, len(ids), dtype=torch.bool)
        mask[0, 2:18] = True
        return dict(input_ids=[ids], image_embeds=img, embeds_cmp_mask=torch.tensor([True]), ids_cmp_mask=mask,
                    patch_positions=torch.tensor([[0.5, 0.5]]))
    ids = [1, 30] + [0] * 16 + [31, 32, 33, 34]
    prefilled = []
    for turn in range(3):
        a = reuse.generate_batch(tok, [req(ids)], reuse_cache=True, **kw)[0]
        prefilled.append(reuse.last_prefill_tokens[0])
        b = full.generate_batch(tok, [req(ids)], **kw)[0]
        assert a["generate_ids"].tolist() == b["generate_ids"].tolist(), turn
        assert relerr(a["last_hidden_states"], b["last_hidden_states"]) < 1e-3
        ids = ids + a["generate_ids"].tolist() + [50 + turn, 51, 52]            # answer + the next user turn
    assert prefilled[0] == 22 and prefilled[1] == 1 + 3 and prefilled[2] == 1 + 3, prefilled   # last answer token + new turn
    # same ids, different image → the fingerprint mismatch at the first image row cuts the reusable prefix to 2 tokens
    img = torch.randn(1, 36, vit_dim, generator=g).to(dev)
    a = reuse.generate_batch(tok, [req(ids)], reuse_cache=True, **kw)[0]
    assert reuse.last_prefill_tokens[0] == len(ids) - 2
    b = full.generate_batch(tok, [req(ids)], **kw)[0]
    assert a["generate_ids"].tolist() == b["generate_ids"].tolist()
