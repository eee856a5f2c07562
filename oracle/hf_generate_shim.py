"""Stand-in for transformers 4.30.2 ``GenerationMixin.generate`` → ``greedy_search`` [ext], bound onto the REFERENCE's own
``LlamaForCausalLM`` so that the reference's ``ContinuousLVLM.generate`` (src/models/mllm/seed_x.py:130-223) can be EXECUTED
here (test infrastructure — build container only; nothing in the product path imports this file).

Why a stand-in: the reference pins transformers 4.30.2 (requirements.txt); this image has 5.15 where ``PreTrainedModel`` no
longer inherits ``GenerationMixin`` (``hasattr(LlamaForCausalLM, 'generate')`` is False), so ``self.llm.generate(...)`` at
seed_x.py:184 has nothing to call. Only the generic loop is restated; every model-specific piece it drives is the reference's
own code, executed: ``prepare_inputs_for_generation`` (modeling_llama_xformer.py:748-779), ``forward`` (:643-746),
``AutoImageTokenGenerationProcessor.__call__`` (generation.py:19-31) through transformers' own ``LogitsProcessorList``.

What 4.30.2 does for the call at seed_x.py:184-189 (``input_ids=…, inputs_embeds=…, output_hidden_states=True,
return_dict_in_generate=True, logits_processor=…, temperature=0.7, num_beams=1, max_new_tokens=…, top_p=0.5, do_sample=False``),
restated step by step (generation/utils.py of that release):
  * ``_prepare_model_inputs``: a decoder-only model given ``inputs_embeds`` next to ``input_ids`` keeps ``input_ids`` for the
    sequence bookkeeping and forwards ``inputs_embeds`` on the first step only — allowed because the model's
    ``prepare_inputs_for_generation`` has an ``inputs_embeds`` parameter (checked by signature inspection, as here);
  * no ``attention_mask`` given → ``torch.ones(inputs_embeds.shape[:2], dtype=long)`` (``_prepare_attention_mask_for_generation``
    for a non-integer input);
  * ``max_length = max_new_tokens + input_ids.shape[-1]``; mode = greedy (``num_beams == 1``, ``do_sample=False``):
    ``temperature`` / ``top_p`` only configure sampling warpers, which greedy search never builds → inert;
  * logits processors: the default list is empty for this configuration, the caller's list is appended;
  * ``eos_token_id`` / ``pad_token_id`` from ``model.config`` (``GenerationConfig.from_model_config``); pad ← eos when pad is None;
  * ``greedy_search``: loop { ``model_inputs = self.prepare_inputs_for_generation(input_ids, **model_kwargs)``; ``outputs =
    self(**model_inputs, return_dict=True, output_attentions=…, output_hidden_states=…)``; ``scores = processors(input_ids,
    outputs.logits[:, -1, :])``; collect ``outputs.hidden_states``; ``next = argmax``; finished rows emit pad; append;
    ``_update_model_kwargs_for_generation`` (past_key_values ← outputs.past_key_values, attention_mask gets a column of
    ones); stop when every row has emitted EOS or ``MaxLengthCriteria`` fires }.
  * returns ``GreedySearchDecoderOnlyOutput(sequences, scores, attentions, hidden_states)``: ``sequences`` includes the prompt
    ids, ``hidden_states`` is one tuple (embeddings + every layer; last entry post-final-norm in this fork, :595-599) per step.
"""
import inspect
import types

import torch


class GreedySearchDecoderOnlyOutput(dict):
    """Attribute- and key-accessible like transformers' ModelOutput."""
    __getattr__ = dict.get


def generate(self, inputs=None, logits_processor=None, **kwargs):
    kwargs = dict(kwargs)
    # -- generation-config keys (everything else is a model kwarg) -------------------------------------------------------
    max_new_tokens = kwargs.pop("max_new_tokens", None)
    assert max_new_tokens is not None, "stand-in: only the max_new_tokens form used at seed_x.py:178 is restated"
    do_sample = kwargs.pop("do_sample", False)
    num_beams = kwargs.pop("num_beams", 1)
    kwargs.pop("temperature", None)            # sampling warpers only (inert under greedy search)
    kwargs.pop("top_p", None)
    assert not do_sample and num_beams == 1, "stand-in: greedy search only (seed_x.py:175-181)"
    output_hidden_states = kwargs.pop("output_hidden_states", self.config.output_hidden_states)
    output_attentions = kwargs.pop("output_attentions", self.config.output_attentions)
    output_scores = kwargs.pop("output_scores", False)
    return_dict_in_generate = kwargs.pop("return_dict_in_generate", False)
    use_cache = kwargs.pop("use_cache", True)                       # GenerationConfig default
    eos_token_id = kwargs.pop("eos_token_id", self.config.eos_token_id)
    pad_token_id = kwargs.pop("pad_token_id", self.config.pad_token_id)
    if pad_token_id is None and eos_token_id is not None:
        pad_token_id = eos_token_id[0] if isinstance(eos_token_id, (list, tuple)) else eos_token_id
    model_kwargs = kwargs

    # -- _prepare_model_inputs ---------------------------------------------------------------------------------------------
    input_ids = model_kwargs.pop("input_ids", None)
    if inputs is not None:
        assert input_ids is None, "`inputs` and `input_ids` were both passed"
        input_ids = inputs
    if "inputs_embeds" in model_kwargs:
        if "inputs_embeds" not in set(inspect.signature(self.prepare_inputs_for_generation).parameters.keys()):
            raise ValueError("You passed `inputs_embeds` to `.generate()`, but the model class doesn't have its forwarding "
                             "implemented.")
        inputs_tensor = model_kwargs["inputs_embeds"]
        if input_ids is None:        # _maybe_initialize_input_ids_for_generation: a bos column (unused by seed_x.py)
            input_ids = torch.full((inputs_tensor.shape[0], 1), self.config.bos_token_id, dtype=torch.long,
                                   device=inputs_tensor.device)
    else:
        inputs_tensor = input_ids
    model_kwargs["output_attentions"] = output_attentions
    model_kwargs["output_hidden_states"] = output_hidden_states
    model_kwargs["use_cache"] = use_cache
    if model_kwargs.get("attention_mask", None) is None:
        if inputs_tensor.dim() == 2 and inputs_tensor.dtype in (torch.int, torch.long) and pad_token_id is not None \
                and (inputs_tensor == pad_token_id).any() and pad_token_id != eos_token_id:
            model_kwargs["attention_mask"] = inputs_tensor.ne(pad_token_id).long()
        else:
            model_kwargs["attention_mask"] = torch.ones(inputs_tensor.shape[:2], dtype=torch.long, device=inputs_tensor.device)
    max_length = max_new_tokens + input_ids.shape[-1]
    from transformers import LogitsProcessorList
    processors = LogitsProcessorList()
    if logits_processor is not None:
        processors.extend(logits_processor)

    # -- greedy_search -----------------------------------------------------------------------------------------------------
    if isinstance(eos_token_id, int):
        eos_token_id = [eos_token_id]
    eos_t = torch.tensor(eos_token_id).to(input_ids.device) if eos_token_id is not None else None
    scores = () if (return_dict_in_generate and output_scores) else None
    decoder_hidden_states = () if (return_dict_in_generate and output_hidden_states) else None
    unfinished = torch.ones(input_ids.shape[0], dtype=torch.long, device=input_ids.device)
    # forward kwargs that greedy_search passes explicitly are not model inputs produced by prepare_inputs_for_generation
    loop_kwargs = {k: v for k, v in model_kwargs.items() if k not in ("output_attentions", "output_hidden_states")}
    while True:
        model_inputs = self.prepare_inputs_for_generation(input_ids, **loop_kwargs)
        outputs = self(**model_inputs, return_dict=True, output_attentions=output_attentions,
                       output_hidden_states=output_hidden_states)
        next_token_logits = outputs.logits[:, -1, :]
        next_tokens_scores = processors(input_ids, next_token_logits)
        if return_dict_in_generate:
            if output_scores:
                scores += (next_tokens_scores,)
            if output_hidden_states:
                decoder_hidden_states += (outputs.hidden_states,)
        next_tokens = torch.argmax(next_tokens_scores, dim=-1)
        if eos_t is not None:
            if pad_token_id is None:
                raise ValueError("If `eos_token_id` is defined, make sure that `pad_token_id` is defined.")
            next_tokens = next_tokens * unfinished + pad_token_id * (1 - unfinished)
        input_ids = torch.cat([input_ids, next_tokens[:, None]], dim=-1)
        # _update_model_kwargs_for_generation
        loop_kwargs["past_key_values"] = outputs.past_key_values
        am = loop_kwargs["attention_mask"]
        loop_kwargs["attention_mask"] = torch.cat([am, am.new_ones((am.shape[0], 1))], dim=-1)
        if eos_t is not None:
            unfinished = unfinished.mul(next_tokens.tile(eos_t.shape[0], 1).ne(eos_t.unsqueeze(1)).prod(dim=0))
        if unfinished.max() == 0 or input_ids.shape[-1] >= max_length:        # MaxLengthCriteria
            break
    if return_dict_in_generate:
        return GreedySearchDecoderOnlyOutput(sequences=input_ids, scores=scores, attentions=None,
                                             hidden_states=decoder_hidden_states)
    return input_ids


def bind(llm):
    """Gives a reference LlamaForCausalLM INSTANCE the 4.30.2-style ``generate``."""
    llm.generate = types.MethodType(generate, llm)
    return llm


# ----------------------------------------------------------------------------------------------------------------------
# the reference's ContinuousLVLM, built from the reference classes around seeded weights
# ----------------------------------------------------------------------------------------------------------------------
class StubTokenizer:
    """Whitespace tokenizer over decimal ids carrying the reference's special tokens at fixed ids:
    ``<img>`` = 400, ``<img_00000>``… = 401…, ``</img>`` = 465 (the 500-entry vocabulary of oracle.weights.MINI_LLM).
    Implements exactly what seed_x.py / generation.py call: ``encode(s, add_special_tokens=False)``,
    ``decode(ids, skip_special_tokens=False)``, ``tokenizer(prompt, return_tensors='pt').input_ids``."""
    eos_token_id = 2
    bos_token_id = 1

    def encode(self, s, add_special_tokens=False):
        import re
        out = [self.bos_token_id] if add_special_tokens else []
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

    def __call__(self, prompt, return_tensors="pt"):
        return types.SimpleNamespace(input_ids=torch.tensor([self.encode(prompt, add_special_tokens=True)]))


def build_reference_lvlm(cfg, sd_llm, sd_agent, vit_dim, in_grid, out_grid, heads):
    """Reference ``ContinuousLVLM`` (seed_x.py:22-46) in fp32 on CPU: reference ``LlamaForCausalLM`` + two reference
    ``Resampler``s (agent_seed_x_i.yaml geometry at mini dims), ``add_patch_pos=True, vit_down=True, mse=True`` as in
    configs/clm_models/agent_seed_x_i.yaml; weights loaded through ``load_state_dict`` with the reference's key names."""
    from transformers import LlamaConfig
    from . import refshim
    mods = refshim.reference_modules()
    from src.models.mllm.seed_x import ContinuousLVLM
    llm = mods["LlamaForCausalLM"](LlamaConfig(**cfg)).eval()
    full = dict(llm.state_dict())
    full.update(sd_llm)
    llm.load_state_dict(full, strict=True)
    bind(llm)
    H = cfg["hidden_size"]
    rin = mods["Resampler"](grid_size=in_grid, embed_dim=H, num_heads=heads, kv_dim=vit_dim)
    rout = mods["Resampler"](grid_size=out_grid, embed_dim=vit_dim, num_heads=heads, kv_dim=H)
    m = ContinuousLVLM(llm, rin, rout, add_patch_pos=True, vit_down=True, mse=True).eval()
    missing, unexpected = m.load_state_dict(sd_agent, strict=False)
    assert not unexpected and all(k.startswith("llm.") for k in missing), (missing, unexpected)
    return m
