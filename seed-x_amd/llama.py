"""Path B — Llama-style decoder on the HIP kernels (prefill on MFMA, single-token decode on the HBM-bound
GEMV path, device-resident greedy loop captured in a HIP graph, optional lock-step batching of independent sequences).

Mirrors ``src/models/mllm/modeling_llama_xformer.py`` (``LlamaForCausalLM``: ``forward`` :643-746 semantics —
logits, past_key_values, hidden_states with the LAST entry = post-final-norm state :595-599 —,
``get_input_embeddings`` :623) with the reference's state-dict key names. Differences by design:
  * KV cache is pre-allocated [layer][seq][head][Tmax][hd] and appended in place by the fused RoPE kernel, instead of
    ``torch.cat`` every step (:215-220)
  * q/k/v and gate/up projections are fused GEMMs (weights concatenated / GLU-packed at load)
  * the 4 NaN/Inf guards (:702-713) and the per-layer ``attention_mask.sum()`` sync (:236) are dropped: prefill and
    multi-token chunks are causal (bottom-right aligned), q_len == 1 attends to the whole cache — same math
  * the generate loop only produces the last position's logits (``forward`` returns all positions like the reference)
  * ``max_batch`` > 1: G independent sequences (separate KV caches / positions) decode in lock step, so the 25.7 GB of
    weights are streamed from HBM once per step for all G tokens (the reference is batch 1 only, seed_x.py:191)
  * ``precise`` (default on; ``SX_LLM_PRECISE=0`` or ``precise=False`` for the plain 16-bit flow): fp32-grade activations — every GEMM
    A operand travels as two 16-bit planes x = hi + lo (the 16-bit checkpoint weights are exact and stream once), q / k / v, RoPE, the
    K cache and attention stay fp32 (csrc/precise.hip; v is stored in 16 bits in fp16 models: the mixed cache). 40 layers at 13B dims:
    logits 5.6e-4 / 5.9e-4 (prefill / decode; 2.9e-5 with the all-fp32 cache) from the reference evaluated in fp32, asserted at 1e-3 (2.3e-3 without;
    the per-site budget is tools/llm_error_budget.py). Measured cost (profiles/r5_llm_precise_ab.log): + 1.8 % of a
    headline step — the LLM phases themselves + 49 % (462 → 688 ms: prefill 2.3x, token step 6.0 → 7.7 ms at 16 sequences).
    MEMORY: the KV cache is fp32 for K and — round 6, fp16 models — 16-bit for V ("mixed" cache, ``kv_v16``; bf16 models and
    ``kv_v16=False`` keep fp32 V) — L·G·heads·Tmax·hd·(4 + 2 | 4) B = 5.0 | 6.7 GB per sequence at 13B dims with
    Tmax = 4096 (pass ``max_cache_len``: the cache is sized from it, 0.63 | 0.84 GB per sequence at 512) — and the decode-tile copy of the
    weights (+ 25.7 GB at 13B) exists for every batch size, not only G >= 5; ``memory_footprint()`` returns the figures before
    anything is allocated, ``_pack`` logs them. The ``past_key_values`` views ``forward`` returns are fp32 in this mode (the reference
    returns the model dtype; they are views of the module's own cache and only meant to be handed back to ``forward``).
    Lock-step batches of 17..32 sequences stay in precise mode (round 6: four operand blocks per weight fragment, 10.9 ms per 32-sequence
    token step); ``SX_LLM_PRECISE32=0`` sends them to the plain 16-bit flow instead (2.0e-3 at 40 layers) — logged once at construction
  * ``comm`` with world > 1: Megatron tensor parallelism (parallel.py) — this rank owns nh/tp heads (their q/k/v rows, KV
    cache and o_proj columns), I/tp FFN rows (gate/up rows, down_proj columns) and Vpad/tp lm_head rows; the fp32
    residual stream is all-reduced after o_proj and down_proj (rank 0's GEMM epilogue adds the residual), the logits are
    all-gathered in front of the replicated greedy rule, so every rank holds identical tokens and loop state
"""
import logging
import math
import os

import torch

from . import ops
from .parallel import Comm, llama_tp_shard


def glu_pack_rows(lin, gate):
    """[I,K] linear rows and [I,K] gate rows → [2I,K] in 32-row groups [16 linear | 16 gate] (sx_gemm glu contract)."""
    I, K = lin.shape
    assert I % 16 == 0
    return torch.cat([lin.view(I // 16, 16, K), gate.view(I // 16, 16, K)], dim=1).reshape(2 * I, K).contiguous()


def merge_peft_lora(sd, lora_alpha=32, adapter="default"):
    """A PeftModel (LoRA) checkpoint of the LLM → the plain LlamaForCausalLM state dict, merged at load time — the optional
    configs/clm_models/llm_seed_x_lora.yaml route (peft.LoraConfig r 32, lora_alpha 32 on q/k/v/o/gate/up/down, modules_to_save = the
    norms): W' = W + (lora_B · lora_A) · lora_alpha / r, exactly peft's Linear.merge() / get_delta_weight()
    (proj/peft/src/peft/tuners/lora.py:779-806); `modules_to_save` copies replace the wrapped module's weight. The product is formed
    in fp32 and cast back to W's dtype. Keys without the `base_model.model.` prefix pass through unchanged."""
    out, lora = {}, {}
    pre = "base_model.model."
    for k, v in sd.items():
        k = k[len(pre):] if k.startswith(pre) else k
        if ".lora_A." in k or ".lora_B." in k:
            mod, rest = k.split(".lora_", 1)
            which, ad = rest.split(".")[0], rest.split(".")[1]
            if ad == adapter or rest.split(".")[1] == "weight":       # `X.lora_A.<adapter>.weight` (or, older peft, `X.lora_A.weight`)
                lora.setdefault(mod, {})[which] = v
            continue
        if ".lora_dropout." in k or ".lora_embedding_" in k:
            continue
        if ".modules_to_save." in k:
            mod, rest = k.split(".modules_to_save.", 1)
            if rest.split(".")[0] == adapter:
                out[mod + "." + rest.split(".", 1)[1]] = v
            continue
        k = k.replace(".original_module.", ".").replace(".base_layer.", ".")
        out.setdefault(k, v)
    for mod, ab in lora.items():
        if "A" not in ab or "B" not in ab:
            raise KeyError(f"merge_peft_lora: {mod} has only one of lora_A / lora_B")
        w = out[mod + ".weight"]
        r = ab["A"].shape[0]
        delta = (ab["B"].float() @ ab["A"].float()) * (float(lora_alpha) / r)
        out[mod + ".weight"] = (w.float() + delta.to(w.device)).to(w.dtype)
    return out


class LlamaConfigLite:
    def __init__(self, hidden_size, intermediate_size, num_hidden_layers, num_attention_heads, vocab_size,
                 rms_norm_eps=1e-5, max_position_embeddings=4096, rope_base=10000.0, **_):
        self.hidden_size, self.intermediate_size = hidden_size, intermediate_size
        self.num_hidden_layers, self.num_attention_heads = num_hidden_layers, num_attention_heads
        self.vocab_size, self.rms_norm_eps = vocab_size, rms_norm_eps
        self.max_position_embeddings, self.rope_base = max_position_embeddings, rope_base


class _Embedding:
    """Callable returned by get_input_embeddings(): ids → fp32 embeddings (seed_x.py:158)."""

    def __init__(self, owner):
        self.owner = owner

    def __call__(self, input_ids):
        ids = input_ids.to(device=self.owner.device, dtype=torch.int32).reshape(-1).contiguous()
        e = ops.embedding(ids, self.owner._P["embed"])
        return e.view(*input_ids.shape, -1)


class CausalLMOutputWithPast(dict):
    """Stand-in for transformers.modeling_outputs.CausalLMOutputWithPast (modeling_llama_xformer.py:740-746): fields by
    attribute (``out.logits``), by key (``out["logits"]``) or by position over the non-None fields (``out[0]``)."""

    def __getattr__(self, k):
        try:
            return dict.__getitem__(self, k)
        except KeyError:
            raise AttributeError(k) from None

    def __getitem__(self, k):
        if isinstance(k, (int, slice)):
            return self.to_tuple()[k]
        return dict.__getitem__(self, k)

    def to_tuple(self):
        return tuple(v for v in self.values() if v is not None)


class LlamaForCausalLM:
    def __init__(self, config, max_cache_len=None, max_batch=1, comm=None, precise=None, kv_v16=None):
        self.config = config if not isinstance(config, dict) else LlamaConfigLite(**config)
        c = self.config
        self.H, self.nh, self.L = c.hidden_size, c.num_attention_heads, c.num_hidden_layers
        self.hd = self.H // self.nh
        self.I, self.V = c.intermediate_size, c.vocab_size
        self.comm = comm or Comm()
        tp = self.tp = self.comm.world
        self.Vpad = (self.V + 64 * tp - 1) // (64 * tp) * (64 * tp)   # 32330 → 32384: lm_head rows padded with zeros
        assert self.nh % tp == 0 and self.I % (16 * tp) == 0, f"tp={tp} does not divide heads {self.nh} / FFN {self.I}"
        self.nh_l, self.I_l, self.V_l = self.nh // tp, self.I // tp, self.Vpad // tp
        self.H_l = self.nh_l * self.hd
        self.Tmax = max_cache_len or c.max_position_embeddings
        self.G = int(max_batch)
        assert 1 <= self.G <= 32, "lock-step batch is limited to 32 sequences (two 16-row operand blocks of sx_gemv)"
        # fp32-grade activations (module docstring). Up to 16 sequences the skinny GEMM's two operand blocks are the hi and lo planes; 17..32
        # sequences (round 6) run four blocks per weight fragment ([2 planes][2 row blocks], gemm_skinny_kernel<.., MB = 4>): the weights
        # still stream once. SX_LLM_PRECISE32=0 keeps round 5's behaviour (plain 16-bit flow above 16 sequences, logged).
        if precise is None:
            want = os.environ.get("SX_LLM_PRECISE", "1") != "0"
            precise = want and (self.G <= 16 or os.environ.get("SX_LLM_PRECISE32", "1") != "0")
            if want and not precise:      # never a silent change of numerics
                logging.getLogger("seedx_amd").warning(
                    "LlamaForCausalLM(max_batch=%d): more than 16 lock-step sequences run the PLAIN 16-bit flow (one rounding per MFMA "
                    "operand, 16-bit KV cache: logits 2.0e-3 from the fp32 reference at 40 layers, asserted at 3e-3) instead of the precise "
                    "mode (1e-3 contract) because SX_LLM_PRECISE32=0.", self.G)
        self.precise = bool(precise)
        # precise mode's "mixed" KV cache (round 6): k fp32, v in the model's 16-bit dtype — three quarters of the fp32 cache's bytes
        # (the cache is a third of what a token step moves at 1.5k tokens of context). It costs parity where the round-5 mode had a
        # factor 35 to spare: 40-layer logits 2.9e-5 → 5.6e-4 prefill / 5.9e-4 decode in fp16 (profiles/r6_fulldepth.log; asserted at
        # 1e-3). Default: ON for fp16 models, OFF for bf16 (a bf16 v carries 8 mantissa bits: 2.3e-3 on the miniature decoder, outside the
        # contract) — decided in _pack once the dtype is known; ``kv_v16=False`` / ``SX_LLM_V16=0`` keep the all-fp32 cache.
        self._kv_v16_arg = kv_v16
        self.kv_v16 = False
        # Decode attention (tools/bench_decode_attention_ab.py, 16 sequences x 40 heads, ms per token of the graph-replayed step):
        # three launches (RoPE + append, split-KV attention, combine) with 8 / 2 / 1 KV splits 6.70 / 6.46 / 6.52; ONE launch
        # (sx_attn_decode_fused, bit-identical) with 8 splits 6.80 — its arrival-counter tail costs more than two graph
        # launches — but with ONE split per head it needs no partials, counter or second pass at all. So: G x heads >= 512
        # workgroups without splitting → fused, 1 split; fewer → three launches with just enough splits for ~1024 workgroups.
        # Both from the GLOBAL head count: tensor-parallel ranks split exactly like one rank (bit-identical per head).
        self.fused_decode_attention = self.G * self.nh >= 512
        self.decode_nsplit = 1 if self.fused_decode_attention else min(8, max(1, -(-1024 // (self.G * self.nh))))
        # precise mode's fp32 decode attention: one workgroup per (head, sequence) walks ALL keys — 4 sequences x 40 heads are 160
        # workgroups of 94 dependent iterations at 1.5k keys (BASELINE config 5: 4 of the token step's 11 ms). Below 512 workgroups the keys
        # of a head are spread over enough splits for ~1024 (+ a combine launch); from the GLOBAL head count, like decode_nsplit.
        self.decode_nsplit_f32 = 1 if self.G * self.nh >= 512 else min(16, max(1, -(-1024 // (self.G * self.nh))))
        self.device, self.dtype = None, torch.float16
        self._sd, self._P = None, None
        self._graph = None
        self.kv_epoch = 0               # bumped whenever the KV cache is reset or written outside generate_batch

    def memory_footprint(self):
        """Bytes this module will hold on its GPU once packed (per rank): 16-bit weights, their decode-tile copy (precise mode: always;
        plain flow: from 5 lock-step sequences), and the KV cache (fp32 in precise mode) — sized from ``max_cache_len``."""
        per_layer = (3 * self.H_l * self.H + self.H * self.H_l + 2 * self.I_l * self.H + self.H * self.I_l) * 2
        w = self.L * per_layer + (self.V + self.V_l) * self.H * 2
        tiles = self.L * per_layer + self.V_l * self.H * 2 if (self.G >= 5 or self.precise) else 0
        kv = self.L * self.G * self.nh_l * self.Tmax * self.hd * ((4 + (2 if self.kv_v16 else 4)) if self.precise else 4)
        return {"weights": w, "decode_tiles": tiles, "kv_cache": kv, "total": w + tiles + kv}

    # ---- reference-compatible plumbing ---------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, torch_dtype=torch.float16, low_cpu_mem_usage=True, peft_adapter=None, **kw):
        """HF directory (config.json + *.bin / *.safetensors shards), like llm_seed_x_i.yaml:1-3. ``peft_adapter``: optional PEFT LoRA
        adapter directory merged into the weights at load."""
        import glob
        import json
        import os
        cfg = json.load(open(os.path.join(pretrained_model_name_or_path, "config.json")))
        m = cls(LlamaConfigLite(**cfg), **kw)
        sd = {}
        for f in sorted(glob.glob(os.path.join(pretrained_model_name_or_path, "*.safetensors"))):
            from safetensors.torch import load_file
            sd.update(load_file(f))
        for f in sorted(glob.glob(os.path.join(pretrained_model_name_or_path, "pytorch_model*.bin"))):
            sd.update(torch.load(f, map_location="cpu"))
        lora_alpha = 32
        if peft_adapter is not None:
            # a PEFT adapter directory (adapter_config.json + adapter_model.safetensors | .bin), what PeftModel.from_pretrained(model_id)
            # reads in src/models/mllm/peft_models.py:106: merged into the base weights at load (merge_peft_lora)
            acfg = json.load(open(os.path.join(peft_adapter, "adapter_config.json")))
            lora_alpha = acfg.get("lora_alpha", 32)
            fs = os.path.join(peft_adapter, "adapter_model.safetensors")
            if os.path.exists(fs):
                from safetensors.torch import load_file
                ad = load_file(fs)
            else:
                ad = torch.load(os.path.join(peft_adapter, "adapter_model.bin"), map_location="cpu")
            # adapter files name the LoRA matrices `….lora_A.weight` (the adapter name is dropped on save)
            sd = {("base_model.model." + k if not k.startswith("base_model.model.") else k): v for k, v in sd.items()}
            sd.update(ad)
        m.load_state_dict(sd, lora_alpha=lora_alpha)
        m.dtype = torch_dtype
        return m

    def expected_keys(self):
        keys = ["model.embed_tokens.weight", "model.norm.weight", "lm_head.weight"]
        for i in range(self.L):
            p = f"model.layers.{i}."
            keys += [p + f"self_attn.{n}.weight" for n in ("q_proj", "k_proj", "v_proj", "o_proj")]
            keys += [p + f"mlp.{n}.weight" for n in ("gate_proj", "up_proj", "down_proj")]
            keys += [p + "input_layernorm.weight", p + "post_attention_layernorm.weight"]
        return keys

    def load_state_dict(self, sd, strict=True, lora_alpha=32):
        if any(".lora_A." in k for k in sd):                 # a PeftModel checkpoint (llm_seed_x_lora.yaml): merge the adapters at load
            sd = merge_peft_lora(sd, lora_alpha=lora_alpha)
        missing = [k for k in self.expected_keys() if k not in sd]
        if missing and strict:
            raise KeyError(f"LlamaForCausalLM: missing keys {missing[:6]} (+{max(0, len(missing) - 6)})")
        self._sd = sd
        self._P = None
        return missing, []

    def to(self, device=None, dtype=None):
        old = (self.device, self.dtype)
        if device is not None:
            self.device = torch.device(device)
        if dtype is not None:
            assert dtype in (torch.float16, torch.bfloat16)
            self.dtype = dtype
        if (self.device, self.dtype) != old:
            if self._P is not None and self._sd is None:
                raise RuntimeError("weights were already packed for %s/%s; reload the state dict to move them" % old)
            self._P = None
        return self

    def eval(self):
        return self

    def get_input_embeddings(self):
        self._pack()
        return _Embedding(self)

    # ---- packing ---------------------------------------------------------------------------------------------------
    def _pack(self):
        if self._P is not None:
            return self._P
        if self._sd is None:
            raise RuntimeError("LlamaForCausalLM: load_state_dict() first")
        if self.device is None or self.device.type != "cuda":
            raise RuntimeError("LlamaForCausalLM runs on the GPU only")
        sd, dev, dt = self._sd, self.device, self.dtype
        want_v16 = self._kv_v16_arg if self._kv_v16_arg is not None else \
            (os.environ.get("SX_LLM_V16", "1") != "0" and dt == torch.float16)
        self.kv_v16 = bool(want_v16) and self.precise and self.hd <= 128 and self.hd % 8 == 0
        fp = self.memory_footprint()
        free = torch.cuda.mem_get_info(dev)[0] + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)   # + the allocator's cached blocks
        logging.getLogger("seedx_amd").info(
            "LlamaForCausalLM._pack (%s, %d sequences, Tmax %d): weights %.1f GB + decode tiles %.1f GB + KV cache %.1f GB (%s)",
            "precise" if self.precise else "plain 16-bit", self.G, self.Tmax, fp["weights"] / 1e9, fp["decode_tiles"] / 1e9,
            fp["kv_cache"] / 1e9, ("k fp32 + v 16-bit" if self.kv_v16 else "fp32") if self.precise else "16-bit")
        if fp["total"] > free:
            raise RuntimeError(
                f"LlamaForCausalLM: {fp['total'] / 1e9:.1f} GB needed ({fp['weights'] / 1e9:.1f} weights + {fp['decode_tiles'] / 1e9:.1f} decode "
                f"tiles + {fp['kv_cache'] / 1e9:.1f} KV cache at max_cache_len {self.Tmax}, {self.G} sequences"
                f"{', fp32 (precise mode)' if self.precise else ''}) but {free / 1e9:.1f} GB are free: lower max_cache_len / max_batch"
                + (", or pass precise=False (16-bit cache, no decode tiles below 5 sequences)" if self.precise else ""))
        if self.tp > 1:     # the captured decode step all-reduces [G, H] and all-gathers [G, Vpad/tp] fp32 (see _decode_step_body)
            self.comm.require_capacity(self.G * max(self.H, self.V_l))
        f32 = lambda t: t.detach().to(dev, torch.float32).contiguous()
        w16 = lambda t: t.detach().to(dev, dt).contiguous()
        P = {"embed": w16(sd["model.embed_tokens.weight"]), "norm": f32(sd["model.norm.weight"]), "layers": []}
        r, tp = self.comm.rank, self.tp
        v0, v1 = r * self.V_l, min((r + 1) * self.V_l, self.V)          # this rank's vocab rows (tail rows stay zero)
        lm = torch.zeros(self.V_l, self.H, dtype=dt, device=dev)
        if v1 > v0:
            lm[:v1 - v0] = sd["lm_head.weight"][v0:v1].detach().to(dev, dt)
        P["lm_head"] = lm

        def tiles(w):
            """Second copy of a weight in the decode-tile layout for the lock-step batched decode (G >= 5 rows take the MFMA
            skinny GEMM, whose wave-wide loads then cover whole contiguous kilobytes: 2.85 → 4.2 TB/s per layer in situ,
            tools/bench_gemv_layout.py). Costs the weights' size again (25.7 GB of 288 GB at 13B); prefill keeps row-major."""
            N, K = w.shape
            # (precise mode: every lock-step batch size runs the MFMA skinny GEMM — the lo plane is its second operand block — so the
            # tiles exist from G = 1: 15.1 → 8 ms per token step at G = 4, BASELINE config 5)
            return ops.pack_decode_tiles(w) if ((self.G >= 5 or self.precise) and N % 32 == 0 and K % 64 == 0 and K >= 256) else None
        P["lm_head_t"] = tiles(lm)
        # RMSNorm folded into the batched decode step (single rank, G >= 5; sx_gemv_args.x16_out / row_ssq_*): the decode-tile copies
        # of the projections that FOLLOW a norm carry that norm's gamma (W' = W · diag(gamma), product rounded once to 16 bits) —
        # wgu of every layer (post_attention_layernorm) and wqkv of layers >= 1 (input_layernorm; layer 0's input comes from the
        # embedding, not from a GEMV). Prefill keeps the row-major, unfolded weights and the norm kernel.
        bal20 = (self.G >= 5 or self.precise) and os.environ.get("SX_GEMV_BAL20", "1") != "0"
        # the consumer adds the producer's per-workgroup sums of squares 64 at a time: the o / down launches (N = H) must have a
        # multiple of 64 workgroups (H = 5120: 256 with 20-row tiles, 320 without)
        # (asked from the library, not re-derived here: sx_gemv's own workgroup count for an N = H launch in that weight layout)
        from . import _lib
        parts = _lib.load().sx_gemv_ssq_parts(self.H, 0, 2 if (bal20 and self.H % 20 == 0 and self.H // 20 == 256) else 1)
        fold = self.G >= 5 and tp == 1 and parts % 64 == 0 and os.environ.get("SX_RMS_FOLD", "1") != "0" \
            and not self.precise      # (SX_RMS_FOLD=0: A/B switch, tools/; the precise mode norms in fp32 with its own kernel)
        for i in range(self.L):
            p = f"model.layers.{i}."
            sh = llama_tp_shard(sd, p, r, tp, self.nh, self.hd)
            qkv = torch.cat([sh["q"].detach(), sh["k"].detach(), sh["v"].detach()], dim=0)
            gu = glu_pack_rows(sh["up"].detach().to(dev, dt), sh["gate"].detach().to(dev, dt))
            P["layers"].append(dict(
                ln1=f32(sd[p + "input_layernorm.weight"]), ln2=f32(sd[p + "post_attention_layernorm.weight"]),
                wqkv=w16(qkv), wo=w16(sh["o"]), wgu=gu, wd=w16(sh["down"])))
            lw = P["layers"][-1]
            for k in ("wqkv", "wo", "wgu", "wd"):
                lw[k + "_t"] = tiles(lw[k])
            # N = 5120 output rows are 320 16-row groups on 256 CUs; as 256 groups of 20 rows every CU streams the same bytes
            # (sx_gemv w_layout 2) — the o and down projections of the 13B geometry
            for k in ("wo", "wd"):
                lw[k + "_t20"] = None
                if bal20 and lw[k + "_t"] is not None and lw[k].shape[0] % 20 == 0 and lw[k].shape[0] // 20 == 256:
                    lw[k + "_t20"] = ops.pack_decode_tiles20(lw[k])
            if fold and lw["wgu_t"] is not None and lw["wqkv_t"] is not None:
                g2 = lw["ln2"][None, :]
                lw["wgu_t"] = tiles(glu_pack_rows((sh["up"].detach().to(dev, torch.float32) * g2).to(dt),
                                                  (sh["gate"].detach().to(dev, torch.float32) * g2).to(dt)))
                if i > 0:
                    lw["wqkv_t"] = tiles((qkv.to(dev, torch.float32) * lw["ln1"][None, :]).to(dt).contiguous())
        # every decode GEMV has the decode-tile weight copy → its 16-bit inputs travel as operand tiles too (_layers_single)
        P["decode_tiled"] = all(lw[k + "_t"] is not None for lw in P["layers"] for k in ("wqkv", "wo", "wgu", "wd")) \
            and (self.nh_l * self.hd) % 32 == 0
        P["rms_fold"] = fold and P["decode_tiled"] and self.H % 32 == 0
        assert P["rms_fold"] or not fold or not any(lw["wgu_t"] is not None for lw in P["layers"]), \
            "folded decode tiles without the tiled decode path"
        # split-K scratch of the skinny GEMM: counters + 8 partial [16, H] blocks (include/seedx_hip.h: sx_gemv_args.workspace)
        P["gemv_ws"] = torch.zeros(16384 + 8 * 16 * ((2 if self.precise else 1) * ((self.G + 15) // 16)) * self.H * 4, dtype=torch.uint8,
                                   device=dev) if P["decode_tiled"] else None
        # precise decode step on the skinny GEMM (operand tiles, two planes): every projection shape must satisfy its MFMA path
        P["precise_tiled"] = all(k % 64 == 0 and k >= 256 for k in (self.H, self.H_l, self.I_l)) and \
            all(n % 32 == 0 for n in (3 * self.H_l, self.H, 2 * self.I_l, self.V_l))
        # RMSNorm fold of the precise decode step (single rank, decode tiles, a multiple of 64 workgroups in the o / down launches —
        # the plain fold's conditions): the residual GEMV writes the two planes of x * gamma_next and the rows' sums of squares, the
        # projection behind the norm scales by rstd. gamma sits on the ACTIVATION here: the checkpoint's weights stay exact.
        P["rms_fold_precise"] = self.precise and P["precise_tiled"] and P["decode_tiled"] and tp == 1 and parts % 64 == 0 \
            and self.H % 32 == 0 and self.I_l % 32 == 0 and os.environ.get("SX_RMS_FOLD", "1") != "0"
        inv = 1.0 / (self.config.rope_base ** (torch.arange(0, self.hd, 2).float() / self.hd))
        fr = torch.outer(torch.arange(self.Tmax).float(), inv)           # [Tmax, hd/2] fp32 (:97-113)
        P["cos"], P["sin"] = fr.cos().to(dev).contiguous(), fr.sin().to(dev).contiguous()
        G = self.G
        P["kc"] = torch.zeros((self.L, G, self.nh_l, self.Tmax, self.hd), dtype=torch.float32 if self.precise else dt,
                              device=dev)                                                          # this rank's heads
        P["vc"] = torch.zeros_like(P["kc"], dtype=dt) if self.kv_v16 else torch.zeros_like(P["kc"])
        # device-resident loop state, one entry per sequence
        P["pos"] = torch.zeros(G, dtype=torch.int32, device=dev)         # position of the next input token
        P["ctx"] = torch.ones(G, dtype=torch.int32, device=dev)          # pos + 1 (keys visible to that token)
        P["step"] = torch.zeros(G, dtype=torch.int32, device=dev)        # index into out_ids / hidden buffer
        P["cur"] = torch.zeros(G, dtype=torch.int32, device=dev)         # current input token id
        P["attn_cnt"] = torch.zeros(G * self.nh_l, dtype=torch.int32, device=dev)   # arrival counters of the fused decode attention
        P["attn_f32_part"] = torch.empty((G, self.nh_l, self.decode_nsplit_f32, self.hd + 2), dtype=torch.float32, device=dev) \
            if (self.precise and self.decode_nsplit_f32 > 1) else None
        self._P = P
        self._sd = None
        self._graph = None
        return P

    # ---- core passes ---------------------------------------------------------------------------------------------------
    def reset(self, seq=None):
        P = self._pack()
        self.kv_epoch += 1              # invalidates ContinuousLVLM's cross-turn prefix bookkeeping
        s = slice(None) if seq is None else slice(seq, seq + 1)
        P["pos"][s] = 0
        P["step"][s] = 0
        P["ctx"][s] = 1                 # invariant: ctx == pos + 1 (keys visible to the token at `pos`)

    def _layers_multi(self, x, Ts, seqs):
        """Ts[i] tokens appended to sequence seqs[i] at its cache position (prefill of several prompts, or the forced
        image-token chunk of several sequences, as ONE pass): every GEMM runs on all sum(Ts) rows at once — the weights
        stream once per pass instead of once per sequence — while RoPE / KV append / causal flash attention stay per
        sequence (batched into single launches when all sequences share T and position). x: fp32 [sum(Ts), H] residual
        stream, rows ordered like seqs. Returns the final residual stream."""
        P, dt, H, nh, hd = self._P, self.dtype, self.H_l, self.nh_l, self.hd     # local heads under tensor parallelism
        comm, lead = self.comm, self.comm.rank == 0
        pr = self.precise
        n = len(seqs)
        pos_all = P["pos"].tolist()                                              # one host read per pass
        pos0 = [pos_all[g] for g in seqs]
        for T, p0 in zip(Ts, pos0):
            assert p0 + T <= self.Tmax, f"sequence {p0 + T} exceeds the KV cache ({self.Tmax})"
        uniform = n > 1 and len(set(Ts)) == 1 and len(set(pos0)) == 1 and list(seqs) == list(range(seqs[0], seqs[0] + n))
        offs = [0]
        for T in Ts:
            offs.append(offs[-1] + T)
        M = offs[-1]
        eps = self.config.rms_norm_eps
        scale = 1.0 / math.sqrt(hd)
        g0 = seqs[0]
        for li, lw in enumerate(P["layers"]):
            kc_l, vc_l = P["kc"][li], P["vc"][li]
            if pr:
                # fp32-grade activations: operand planes into a_planes = 2 GEMMs with fp32 outputs, fp32 RoPE / cache / attention
                h, _ = ops.rmsnorm_planes(x, lw["ln1"], eps, dt)
                qkv = ops.gemm(h, lw["wqkv"], a_planes=2, out_dtype=torch.float32)    # [M, 3H] fp32
                if uniform:
                    ops.rope_kv_append_f32(qkv, kc_l[g0:g0 + n], vc_l[g0:g0 + n], P["cos"], P["sin"], P["pos"][g0:g0 + n], n, Ts[0],
                                           nh, hd, dt)
                    att = ops.attention_f32(qkv, kc_l[g0:g0 + n], vc_l[g0:g0 + n], P["pos"][g0:g0 + n], n, Ts[0], nh, hd, scale, dt)
                else:
                    att = torch.empty((M, 2 * H), dtype=dt, device=x.device)
                    for i, g in enumerate(seqs):
                        rows = qkv[offs[i]:offs[i + 1]]
                        ops.rope_kv_append_f32(rows, kc_l[g:g + 1], vc_l[g:g + 1], P["cos"], P["sin"], P["pos"][g:g + 1], 1, Ts[i], nh,
                                               hd, dt)
                        att[offs[i]:offs[i + 1]] = ops.attention_f32(rows, kc_l[g:g + 1], vc_l[g:g + 1], P["pos"][g:g + 1], 1, Ts[i],
                                                                     nh, hd, scale, dt)
                x = comm.all_reduce(ops.gemm(att, lw["wo"], a_planes=2, residual=x if lead else None, out_dtype=torch.float32))
                h, _ = ops.rmsnorm_planes(x, lw["ln2"], eps, dt)
                g_ = ops.split16(ops.gemm(h, lw["wgu"], a_planes=2, act="silu", glu=True, out_dtype=torch.float32), dt)
                x = comm.all_reduce(ops.gemm(g_, lw["wd"], a_planes=2, residual=x if lead else None, out_dtype=torch.float32))
                continue
            h = ops.rmsnorm(x, lw["ln1"], eps, dt)
            qkv = ops.gemm(h, lw["wqkv"])                                             # [M, 3H]
            if uniform:
                T, Tk = Ts[0], pos0[0] + Ts[0]
                ops.rope_kv_append_b(qkv, kc_l[g0:g0 + n], vc_l[g0:g0 + n], P["cos"], P["sin"], P["pos"][g0:g0 + n], n, T,
                                     nh, hd)
                q4 = qkv.view(n, T, 3, nh, hd)[:, :, 0]
                k4 = kc_l[g0:g0 + n, :, :Tk].permute(0, 2, 1, 3)                      # [n, Tk, nh, hd] views of the caches
                v4 = vc_l[g0:g0 + n, :, :Tk].permute(0, 2, 1, 3)
                att = ops.attention(q4, k4, v4, scale, causal=True).view(M, H)
            else:
                att = torch.empty((M, H), dtype=dt, device=x.device)
                for i, g in enumerate(seqs):
                    T, Tk = Ts[i], pos0[i] + Ts[i]
                    rows = qkv[offs[i]:offs[i + 1]]
                    ops.rope_kv_append(rows, kc_l[g], vc_l[g], P["cos"], P["sin"], P["pos"][g:g + 1], nh, hd)
                    q4 = rows.view(1, T, 3, nh, hd)[:, :, 0]
                    k4 = kc_l[g][:, :Tk].permute(1, 0, 2).unsqueeze(0)                # [1, Tk, nh, hd] view of the cache
                    v4 = vc_l[g][:, :Tk].permute(1, 0, 2).unsqueeze(0)
                    ops.attention(q4, k4, v4, scale, causal=True, out=att[offs[i]:offs[i + 1]].view(1, T, H))
            x = comm.all_reduce(ops.gemm(att, lw["wo"], residual=x if lead else None, out_dtype=torch.float32))
            h = ops.rmsnorm(x, lw["ln2"], eps, dt)
            g_ = ops.gemm(h, lw["wgu"], act="silu", glu=True)                         # silu(gate) * up, [M, I/tp]
            x = comm.all_reduce(ops.gemm(g_, lw["wd"], residual=x if lead else None, out_dtype=torch.float32))
        if uniform:
            ops.add_i32(P["pos"][g0:g0 + n], Ts[0])
            ops.add_i32(P["ctx"][g0:g0 + n], Ts[0])
        else:
            for i, g in enumerate(seqs):
                ops.add_i32(P["pos"][g:g + 1], Ts[i])
                ops.add_i32(P["ctx"][g:g + 1], Ts[i])
        return x

    def _layers_single(self, x):
        """One token of EVERY sequence (x: fp32 [G, H]) at the device-resident positions: weight-streaming GEMVs with
        M = G rows + split-KV decode attention per sequence. No host reads → graph-capturable."""
        P, dt, H, nh, hd, G = self._P, self.dtype, self.H_l, self.nh_l, self.hd, self.G
        comm, lead = self.comm, self.comm.rank == 0
        eps = self.config.rms_norm_eps
        scale = 1.0 / math.sqrt(hd)
        # G >= 5 sequences run the MFMA skinny GEMM: its 16-bit inputs then travel as operand tiles (ops.Tiled16) from the
        # kernel that produces them (norm, attention combine, GLU epilogue) — one contiguous 1-KB load per operand instead of
        # 16 rows that all sit on the same L2 channel — and the down projection may split K over workgroups (workspace)
        tl, ws = P["decode_tiled"] and G >= 5, P["gemv_ws"]
        if self.precise:
            return self._layers_single_precise(x)
        # folded RMSNorm: the residual GEMVs (o, down) also emit the new residual stream as 16-bit operand tiles (x16) and its rows'
        # sums of squares (ssq); the projection behind the norm reads x16 with gamma-folded weights and scales by rstd — no norm launch
        fold = tl and P["rms_fold"]
        x16 = ssq = None
        nl = len(P["layers"])
        for li, lw in enumerate(P["layers"]):
            if fold and li > 0:
                qkv = ops.gemv(x16, lw["wqkv"], w_tiles=lw["wqkv_t"], ssq_in=(ssq, self.H, eps))
            else:
                h = ops.rmsnorm(x, lw["ln1"], eps, dt, tiled=tl)
                qkv = ops.gemv(h, lw["wqkv"], w_tiles=lw["wqkv_t"])                   # [G, 3H]
            if self.fused_decode_attention and hd % 16 == 0:
                # RoPE + KV append + split-KV attention + combine as one launch (bit-identical to the three-kernel form below)
                att = ops.attn_decode_fused(qkv, P["kc"][li], P["vc"][li], P["pos"], P["cos"], P["sin"], scale, nh, hd,
                                            P["attn_cnt"], nsplit=self.decode_nsplit, out_tiled=tl)
            else:
                ops.rope_kv_append_b(qkv, P["kc"][li], P["vc"][li], P["cos"], P["sin"], P["pos"], G, 1, nh, hd)
                q = qkv[:, :H].unflatten(1, (nh, hd))                                # strided view into qkv: no copy
                att = ops.attn_decode_b(q, P["kc"][li], P["vc"][li], P["ctx"], scale, nsplit=self.decode_nsplit, out_tiled=tl)
            if fold:
                x, x16, ssq = ops.gemv(att, lw["wo"], residual=x, out_dtype=torch.float32, w_tiles=lw["wo_t"], workspace=ws,
                                       emit_norm=True, w_tiles20=lw["wo_t20"])
                g = ops.gemv(x16, lw["wgu"], act="silu", glu=True, w_tiles=lw["wgu_t"], y_tiled=True, ssq_in=(ssq, self.H, eps))
                if li + 1 < nl:
                    x, x16, ssq = ops.gemv(g, lw["wd"], residual=x, out_dtype=torch.float32, w_tiles=lw["wd_t"], workspace=ws,
                                           emit_norm=True, w_tiles20=lw["wd_t20"])
                else:                        # the final norm needs fp32 states (hidden-state output): its own launch
                    x = ops.gemv(g, lw["wd"], residual=x, out_dtype=torch.float32, w_tiles=lw["wd_t"], workspace=ws,
                                 w_tiles20=lw["wd_t20"])
                continue
            x = comm.all_reduce(ops.gemv(att, lw["wo"], residual=x if lead else None, out_dtype=torch.float32,
                                         w_tiles=lw["wo_t"], workspace=ws, w_tiles20=lw["wo_t20"] if tl else None))
            h = ops.rmsnorm(x, lw["ln2"], eps, dt, tiled=tl)
            g = ops.gemv(h, lw["wgu"], act="silu", glu=True, w_tiles=lw["wgu_t"], y_tiled=tl)
            x = comm.all_reduce(ops.gemv(g, lw["wd"], residual=x if lead else None, out_dtype=torch.float32,
                                         w_tiles=lw["wd_t"], workspace=ws, w_tiles20=lw["wd_t20"] if tl else None))
        ops.add_i32(P["pos"], 1)
        ops.add_i32(P["ctx"], 1)
        return x

    def _layers_single_precise(self, x):
        """_layers_single with fp32-grade activations: the x operand of every skinny GEMM is a two-block Tiled16 (hi plane, lo plane;
        sx_gemv x_planes = 2 — each weight fragment feeds two MFMAs, the weights stream once), its outputs are fp32; RoPE, the KV
        cache and attention are fp32 (sx_rope_kv_append_f32, sx_attention_f32 at T = 1). Shapes outside the skinny GEMM's MFMA path
        (miniature test dims) take the a_planes = 2 GEMM instead. No host reads → graph-capturable."""
        P, dt, nh, hd, G = self._P, self.dtype, self.nh_l, self.hd, self.G
        comm, lead = self.comm, self.comm.rank == 0
        eps, scale = self.config.rms_norm_eps, 1.0 / math.sqrt(hd)
        tl, ws, f32 = P["precise_tiled"], P["gemv_ws"], torch.float32
        dtl = tl and P["decode_tiled"]                       # decode-tile weight copies (built for every G in precise mode)

        def lin(xp, lw, k, **kw):
            if tl:
                return ops.gemv(xp, lw[k], w_tiles=lw[k + "_t"] if dtl else None, workspace=ws if dtl else None,
                                w_tiles20=lw.get(k + "_t20") if dtl else None, out_dtype=f32, **kw)
            return ops.gemm(xp, lw[k], a_planes=2, out_dtype=f32, **kw)
        fold = P["rms_fold_precise"]
        fuse_rope = hd == 128 and os.environ.get("SX_LLM_FUSE_ROPE", "1") != "0"
        x16 = ssq = None
        nl = len(P["layers"])
        for li, lw in enumerate(P["layers"]):
            if fold and li > 0:
                qkv = lin(x16, lw, "wqkv", ssq_in=(ssq, self.H, eps))                 # x16 = planes of x * ln1 (written by the down GEMV)
            else:
                h, _ = ops.rmsnorm_planes(x, lw["ln1"], eps, dt, tiled=tl)
                qkv = lin(h, lw, "wqkv")                                              # [G, 3H] fp32
            if fuse_rope:           # RoPE + KV append inside the attention launch (one graph node per layer instead of two)
                att = ops.attention_f32(qkv, P["kc"][li], P["vc"][li], P["pos"], G, 1, nh, hd, scale, dt, tiled=tl,
                                        nsplit=self.decode_nsplit_f32, scratch=P["attn_f32_part"], rope=(P["cos"], P["sin"]))
            else:
                ops.rope_kv_append_f32(qkv, P["kc"][li], P["vc"][li], P["cos"], P["sin"], P["pos"], G, 1, nh, hd, dt)
                att = ops.attention_f32(qkv, P["kc"][li], P["vc"][li], P["pos"], G, 1, nh, hd, scale, dt, tiled=tl,
                                        nsplit=self.decode_nsplit_f32, scratch=P["attn_f32_part"])
            if fold:
                # residual GEMV: fp32 x, the planes of x * gamma of the NEXT norm, the rows' sums of squares; GLU epilogue: planes directly
                x, x16, ssq = lin(att, lw, "wo", residual=x, emit_norm=True, planes_out=True, norm_gamma=lw["ln2"])
                g = ops.gemv(x16, lw["wgu"], act="silu", glu=True, w_tiles=lw["wgu_t"], y_tiled=True, planes_out=True,
                             ssq_in=(ssq, self.H, eps))
                if li + 1 < nl:
                    x, x16, ssq = lin(g, lw, "wd", residual=x, emit_norm=True, planes_out=True, norm_gamma=P["layers"][li + 1]["ln1"])
                else:                        # the final norm wants fp32 states: its own launch (rmsnorm_planes in _decode_step_body)
                    x = lin(g, lw, "wd", residual=x)
                continue
            x = comm.all_reduce(lin(att, lw, "wo", residual=x if lead else None))
            h, _ = ops.rmsnorm_planes(x, lw["ln2"], eps, dt, tiled=tl)
            if tl:                           # SiLU-GLU epilogue writes the two planes of its result itself (no sx_split16 launch)
                g = ops.gemv(h, lw["wgu"], act="silu", glu=True, w_tiles=lw["wgu_t"] if dtl else None, y_tiled=True, planes_out=True)
            else:
                g = ops.split16(lin(h, lw, "wgu", act="silu", glu=True), dt)
            x = comm.all_reduce(lin(g, lw, "wd", residual=x if lead else None))
        ops.add_i32(P["pos"], 1)
        ops.add_i32(P["ctx"], 1)
        return x

    def _final_norm(self, x):
        """model.norm (modeling_llama_xformer.py:595) → fp32 hidden states."""
        if self.precise:
            return ops.rmsnorm_planes(x.contiguous(), self._P["norm"], self.config.rms_norm_eps, self.dtype, want_f32=True,
                                      want_planes=False)[1]
        return ops.rmsnorm(x, self._P["norm"], self.config.rms_norm_eps, torch.float32)

    def _lm_head(self, hn_rows):
        """lm_head (:707) on fp32 post-norm rows → fp32 logits [rows, Vpad / tp]."""
        P = self._P
        if self.precise:
            return ops.linear_planes(hn_rows.contiguous(), P["lm_head"], w_tiles=P["lm_head_t"], out_dtype=torch.float32)
        return ops.linear(ops.cast(hn_rows.contiguous(), self.dtype), P["lm_head"], out_dtype=torch.float32)

    def forward_embeds(self, inputs_embeds, need_logits=True, seq=0):
        """inputs_embeds: fp32 [T, H] on the GPU, appended to sequence `seq` at its current cache position.
        Returns (logits fp32 [Vpad] of the LAST position or None, final-norm hidden states fp32 [T, H])."""
        P = self._pack()
        self.kv_epoch += 1              # a write outside generate_batch: cached-prefix records no longer describe the cache
        x = inputs_embeds.to(device=self.device, dtype=torch.float32).contiguous()
        T = x.shape[0]
        if T == 1 and self.G == 1:
            x = self._layers_single(x)
        else:
            x = self._layers_multi(x, [T], [seq])
        hn = self._final_norm(x)                                                     # :595
        logits = None
        if need_logits:
            logits = self._lm_head(hn[-1:])[0]
            if self.tp > 1:
                logits = self.comm.all_gather(logits).reshape(-1)                     # [tp, V/tp] → [Vpad], vocab order
        return logits, hn

    def forward_embeds_batch(self, xs, seqs, need_logits=True):
        """xs[i]: fp32 [T_i, H] appended to sequence seqs[i]; all sequences in one pass (see _layers_multi).
        Returns (logits fp32 [n, Vpad] of each sequence's LAST position or None, list of final-norm states [T_i, H])."""
        P = self._pack()
        Ts = [int(x.shape[0]) for x in xs]
        x = torch.cat([x.to(device=self.device, dtype=torch.float32) for x in xs], dim=0).contiguous()   # plumbing
        x = self._layers_multi(x, Ts, list(seqs))
        hn = self._final_norm(x)
        ends = torch.tensor([sum(Ts[:i + 1]) - 1 for i in range(len(Ts))], device=self.device)
        logits = None
        if need_logits:
            logits = self._lm_head(hn[ends])                                           # [n, Vpad / tp]
            if self.tp > 1:
                logits = self.comm.all_gather(logits).permute(1, 0, 2).reshape(len(Ts), self.Vpad).contiguous()
        return logits, list(torch.split(hn, Ts, dim=0))

    def set_position(self, seq, pos):
        """Truncate / rewind sequence `seq`'s KV cache to `pos` tokens (cross-turn prefix reuse)."""
        P = self._pack()
        assert 0 <= pos <= self.Tmax
        P["pos"][seq] = pos
        P["ctx"][seq] = pos + 1

    # reference-style entry (prefill + cached steps through inputs_embeds / input_ids), batch 1
    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None,
                labels=None, use_cache=True, output_attentions=None, output_hidden_states=None, return_dict=True,
                logits_positions="all"):
        """The reference forward's inference contract (modeling_llama_xformer.py:643-746), batch 1 (sequence 0).
        Returns a ``CausalLMOutputWithPast`` (attribute / key / index access like transformers' ModelOutput) with
          * ``logits`` [1, T, V] fp32 for EVERY position (:707; ``logits_positions="last"`` → [1, 1, V], what the greedy loop needs),
          * ``past_key_values``: per layer (k, v) VIEWS [1, heads, T_total, hd] of the module's pre-allocated cache (:217-220 builds
            them with torch.cat). Passing any non-None value back continues from the internal cache — its length must equal the
            cache position; ``None`` starts a new sequence,
          * ``hidden_states`` (if requested): a tuple of L + 1 entries like the reference's (:580-599) whose LAST entry is the
            post-final-norm state [1, T, H]; the intermediate entries are None (never materialised; the path only reads [-1],
            seed_x.py:98,196).
        ``attention_mask`` is accepted and ignored exactly as far as the reference ignores it (padding is never honoured, :236;
        prefill is causal, a q_len == 1 step sees the whole cache); ``position_ids`` must be the default arange (what
        prepare_inputs_for_generation :759-765 produces for an all-ones mask). Training inputs (``labels``) are out of scope."""
        if labels is not None:
            raise NotImplementedError("LlamaForCausalLM.forward: the loss path (labels) is training-side and not built")
        if attention_mask is not None and not bool(torch.as_tensor(attention_mask).to(torch.bool).all()):
            import warnings       # the reference does the same silently (:236 only asks whether the mask is all zero); say it once
            warnings.warn("LlamaForCausalLM.forward: attention_mask has zeros (padding) — like the reference's xformers path "
                          "(modeling_llama_xformer.py:232-237) padding is NOT masked: every position attends causally", stacklevel=2)
        if output_attentions:
            raise NotImplementedError("LlamaForCausalLM.forward: attention probabilities are never materialised (flash attention)")
        P = self._pack()
        if past_key_values is None:
            self.reset(0)
        else:
            held = int(P["pos"][0].item())
            if isinstance(past_key_values, (tuple, list)) and len(past_key_values) and isinstance(past_key_values[0], (tuple, list)):
                got = int(past_key_values[0][0].shape[2])
                assert got == held, f"past_key_values holds {got} tokens but the module's cache is at position {held}"
        if inputs_embeds is None:
            inputs_embeds = self.get_input_embeddings()(input_ids)
        x = inputs_embeds.reshape(-1, self.H)
        T = x.shape[0]
        if position_ids is not None:
            p0 = int(P["pos"][0].item())
            want = torch.arange(p0, p0 + T)
            assert torch.equal(position_ids.reshape(-1).cpu().long(), want), "only the default (arange) position_ids are supported"
        if logits_positions == "last":
            logits, hn = self.forward_embeds(x, seq=0)
            logits = logits[: self.V].view(1, 1, -1)
        else:
            _, hn = self.forward_embeds(x, need_logits=False, seq=0)
            logits = self._lm_head(hn)                                                                           # [T, Vpad / tp]
            if self.tp > 1:
                logits = self.comm.all_gather(logits).permute(1, 0, 2).reshape(T, self.Vpad)
            logits = logits[:, : self.V].unsqueeze(0)
        Tk = int(P["pos"][0].item())
        pkv = tuple((P["kc"][li][0][:, :Tk].unsqueeze(0), P["vc"][li][0][:, :Tk].unsqueeze(0)) for li in range(self.L)) \
            if use_cache else None
        hs = ((None,) * self.L + (hn.view(1, -1, self.H),)) if output_hidden_states else None
        out = CausalLMOutputWithPast(loss=None, logits=logits, past_key_values=pkv, hidden_states=hs, attentions=None)
        return out if return_dict else out.to_tuple()

    __call__ = forward

    # ---- device-resident greedy decode step (all sequences in lock step) -----------------------------------------------
    def _decode_step_body(self, img_ids_dev, out_ids, hid_buf):
        """cur[G] → embedding → 40 layers → final norm → lm_head → logits rule + argmax → cur[G]. Records the post-norm
        hidden state of each INPUT token at hid_buf[g, step[g]] (what seed_x.py:196 collects) and the new id at
        out_ids[g, step[g]]; then step += 1. Everything stays on the device."""
        P = self._P
        x = ops.embedding(P["cur"], P["embed"])                                        # [G, H] fp32
        x = self._layers_single(x)
        if self.precise:
            hp, hn = ops.rmsnorm_planes(x, P["norm"], self.config.rms_norm_eps, self.dtype, tiled=P["precise_tiled"], want_f32=True)
            ops.scatter_rows_step(hn, P["step"], hid_buf)
            if P["precise_tiled"]:
                logits = ops.gemv(hp, P["lm_head"], out_dtype=torch.float32, w_tiles=P["lm_head_t"])
            else:
                logits = ops.gemm(hp, P["lm_head"], a_planes=2, out_dtype=torch.float32)
        else:
            hn = ops.rmsnorm(x, P["norm"], self.config.rms_norm_eps, torch.float32)
            ops.scatter_rows_step(hn, P["step"], hid_buf)
            logits = ops.gemv(ops.cast(hn, self.dtype), P["lm_head"], out_dtype=torch.float32,
                              w_tiles=P["lm_head_t"])                                  # [G, Vpad / tp]
        if self.tp > 1:
            logits = self.comm.all_gather(logits).permute(1, 0, 2).reshape(self.G, self.Vpad).contiguous()
        ops.greedy_next_b(logits, self.V, img_ids_dev, P["cur"], out_ids, P["step"])
        ops.add_i32(P["step"], 1)

    def decode_step(self, img_ids_dev, out_ids, hid_buf, use_graph=True):
        """out_ids: int32 [G, rows]; hid_buf: fp32 [G, rows, H]."""
        self._pack()
        assert out_ids.shape[0] == self.G and hid_buf.shape[0] == self.G and hid_buf.shape[1] == out_ids.shape[1]
        if not use_graph or not self.comm.graph_safe:
            self._decode_step_body(img_ids_dev, out_ids, hid_buf)
            return
        key = (img_ids_dev.data_ptr(), out_ids.data_ptr(), hid_buf.data_ptr(), tuple(out_ids.shape))
        if self._graph is None or self._graph[0] != key:
            # warm-up on a side stream (allocator / lazy module load), then capture one token step
            P = self._P
            snap = {k: P[k].clone() for k in ("pos", "ctx", "step", "cur")}
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                self._decode_step_body(img_ids_dev, out_ids, hid_buf)
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            for k, v in snap.items():
                P[k].copy_(v)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._decode_step_body(img_ids_dev, out_ids, hid_buf)
            for k, v in snap.items():
                P[k].copy_(v)
            self._graph = (key, g)
        self._graph[1].replay()
