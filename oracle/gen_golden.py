"""Generate the golden fixtures in tests/golden/ from the REFERENCE's own modules (build container only).

The reference ships no golden vectors, so the fixtures are produced here by importing
``/root/reference/src/models/*`` on CPU (through oracle/refshim.py), loading the seeded weights of oracle/weights.py
(regenerated deterministically from the seed on any machine with this torch build) and recording input/output
tensors at "mini" dimensions. They pin (a) oracle/restated.py in the CPU suite and (b) the HIP path in the GPU suite
on machines where /root/reference does not exist.

    python -m oracle.gen_golden          # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import refshim, weights  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def _save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name), **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                                     for k, v in arrs.items()})
    print("wrote", name, {k: tuple(np.asarray(v.detach().cpu() if torch.is_tensor(v) else v).shape) for k, v in arrs.items()})


def main():
    mods = refshim.reference_modules()
    g = torch.Generator().manual_seed(2024)
    with torch.no_grad():
        # ---- path A: VisionTransformerWithAttnPool (qwen_visual.py) -------------------------------------
        for tag, cfg in (("vit_hd128", weights.MINI_VIT), ("vit_hd104", weights.MINI_VIT_104)):
            m = mods["VisionTransformerWithAttnPool"](**cfg).eval()
            m.load_state_dict(weights.vit_sd(cfg), strict=True)
            x = torch.randn(2, 3, cfg["image_size"], cfg["image_size"], generator=g)
            _save(f"{tag}.npz", x=x, y=m(x).float())
        # ---- path B: LlamaForCausalLM forward (prefill + one cached step) + logits processor ----------------
        from transformers import LlamaConfig
        cfg = weights.MINI_LLM
        sd = weights.llama_sd(cfg)
        llm = mods["LlamaForCausalLM"](LlamaConfig(**cfg)).eval()
        full = dict(llm.state_dict())
        full.update(sd)
        llm.load_state_dict(full, strict=True)
        x = torch.randn(1, 23, cfg["hidden_size"], generator=g) * 0.5
        o = llm(inputs_embeds=x, attention_mask=torch.ones(1, 23, dtype=torch.long), use_cache=True,
                output_hidden_states=True, return_dict=True)
        tok = torch.tensor([[17]])
        o2 = llm(input_ids=tok, attention_mask=torch.ones(1, 24, dtype=torch.long), past_key_values=o.past_key_values,
                 use_cache=True, output_hidden_states=True, return_dict=True)
        _save("llama_mini.npz", x=x, logits=o.logits, hidden=o.hidden_states[-1], tok=tok, logits2=o2.logits,
              hidden2=o2.hidden_states[-1])
        ids = list(range(400, 466))

        class Tok:
            def encode(self, s, add_special_tokens=False):
                return ids
        proc = mods["AutoImageTokenGenerationProcessor"](Tok(), num_img_gen_tokens=64)
        lasts, scores_in, scores_out = [5, 400, 433, 464, 465], [], []
        for last in lasts:
            sc = torch.randn(1, 500, generator=g) - 3.0
            scores_in.append(sc.clone())
            scores_out.append(proc(torch.tensor([[1, 2, last]]), sc.clone()))
        _save("logits_rule.npz", last=np.array(lasts), scores_in=torch.cat(scores_in), scores_out=torch.cat(scores_out))
        # ---- Resampler (LLM side) and ResamplerXLV2 (path C head) --------------------------------------------
        r = mods["Resampler"](grid_size=4, embed_dim=320, num_heads=2, kv_dim=256).eval()
        r.load_state_dict(weights.resampler_sd(weights._g(7), "", 4, 320, 256), strict=True)
        x = torch.randn(2, 16, 256, generator=g)
        _save("resampler_mini.npz", x=x, y=r(x))
        cfgx = weights.MINI_XLV2
        m = mods["ResamplerXLV2"](normalize=False, **cfgx).eval()
        m.load_state_dict(weights.xlv2_sd(cfgx, pre=""), strict=True)
        x = torch.randn(2, 24, cfgx["embedding_dim"], generator=g)
        pe, pooled = m(x)
        _save("xlv2_mini.npz", x=x, prompt=pe, pooled=pooled)


if __name__ == "__main__":
    if not refshim.available():
        raise SystemExit("/root/reference is not available: golden fixtures can only be generated in the build container")
    main()
