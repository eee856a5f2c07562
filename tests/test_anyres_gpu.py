"""BASELINE config 5 style at mini dimensions: several images, several any-res crops per image, a multi-turn prompt
(layout of src/data/sft_clm.py:229-276: `[INST] … [/INST]\\n answer \\n[INST] …`), a crop that is present in
`image_embeds` but NOT referenced by the prompt (embeds_cmp_mask False), text-only continuation with an EOS stop.
HIP ContinuousLVLM.generate vs the CPU oracle (teacher-forced comparison as in test_models_gpu)."""
import pytest
import torch

from oracle import restated, weights
from tests.test_models_gpu import StubTokenizer, _build_lvlm, relerr

pytestmark = pytest.mark.gpu


def test_multi_image_multi_turn_generate(dev):
    dtype = torch.float16
    cfg = weights.MINI_LLM
    vit_dim = 128
    sd_llm = weights.llama_sd(cfg)
    sd_agent = weights.agent_sd(cfg, vit_dim, in_grid=4, out_grid=4)
    g = torch.Generator().manual_seed(11)
    n_crops = 5                                               # image 1: 2 tiles + global; image 2: 1 tile + global
    image_embeds = torch.randn(n_crops, 36, vit_dim, generator=g)
    emask = torch.tensor([True, True, True, False, True])     # crop 3 is not referenced by the prompt
    ppos = torch.tensor([[0.0, 0.0], [0.5, 0.0], [0.5, 0.5], [0.0, 0.0], [0.5, 0.5]])
    slot = [0] * 16
    ids = [1, 30, 31] + slot + [32] + slot + [33] + slot + [40, 41, 42, 43] + [50, 51] + slot + [60, 61, 62]   # 2 turns
    mask = torch.zeros(1, len(ids), dtype=torch.bool)
    pos = 3
    for k in range(4):                                         # the 4 referenced crops, in order
        mask[0, pos:pos + 16] = True
        pos += 16 + (1 if k < 2 else (6 if k == 2 else 0))
    assert int(mask.sum()) == 64
    img_ids = list(range(400, 466))
    rc = {"in_heads": 2, "out_heads": 2}
    agent = _build_lvlm(dev, dtype, sd_llm, sd_agent, cfg, vit_dim)
    tok = StubTokenizer()
    out = agent.generate(tok, input_ids=[ids], image_embeds=image_embeds.to(dev), embeds_cmp_mask=emask,
                         ids_cmp_mask=mask, patch_positions=ppos, max_new_tokens=12, num_img_gen_tokens=16,
                         eos_token_id=None)
    new = out["generate_ids"].tolist()
    trace = []
    ref = restated.lvlm_generate(sd_llm, sd_agent, cfg, rc, ids, image_embeds, emask, mask, ppos, img_ids, 400, 465, 12,
                                 16, None, None, new, trace)
    for step, (o_arg, forced, gap, std) in enumerate(trace):
        assert o_arg == forced or gap < 0.02 * max(std, 1.0), (step, o_arg, forced, gap, std)
    assert relerr(out["last_hidden_states"], ref["last_hidden"]) < 4e-3
    # EOS stop: declare the 3rd generated id to be EOS → generation must end right there, like HF greedy_search
    tok.eos_token_id = new[2]
    out2 = agent.generate(tok, input_ids=[ids], image_embeds=image_embeds.to(dev), embeds_cmp_mask=emask,
                          ids_cmp_mask=mask, patch_positions=ppos, max_new_tokens=12, num_img_gen_tokens=16)
    assert out2["generate_ids"].tolist() == new[:3]
    assert not out2["has_img_output"] or 465 in new[:3]
