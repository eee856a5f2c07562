"""CPU (fp32, plain torch) restatement of the SEED-X hot path — the parity oracle ("port").

TEST INFRASTRUCTURE: not imported by the product package. Every function follows a reference file:line and is
driven by a state dict that uses the REFERENCE's parameter names, so the same seeded weights feed (a) the real
reference modules in the build container (tests/test_oracle_vs_reference.py, oracle/gen_golden.py), (b) this
restatement, and (c) the HIP path.

Paths A (ViT, qwen_visual.py), B (Llama + ContinuousLVLM.generate, modeling_llama_xformer.py / seed_x.py /
generation.py) and the head of C (ResamplerXLV2, resampler.py) live here; the diffusers UNet / scheduler / CFG
loops are in restated_unet.py.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------------------
# shared pieces
# ---------------------------------------------------------------------------------------------------------
def get_abs_pos(abs_pos, tgt_len):
    """qwen_visual.py:24-40 — bicubic (align_corners=False) resize of a square pos table, fp32."""
    src = int(math.sqrt(abs_pos.size(0)))
    tgt = int(math.sqrt(tgt_len))
    if src == tgt:
        return abs_pos
    return F.interpolate(abs_pos.float().reshape(1, src, src, -1).permute(0, 3, 1, 2), size=(tgt, tgt), mode="bicubic",
                         align_corners=False).permute(0, 2, 3, 1).flatten(0, 2).to(abs_pos.dtype)


def sincos_2d(embed_dim, grid_size):
    """qwen_visual.py:44-91 (get_2d_sincos_pos_embed, w goes first in the meshgrid)."""
    gh = np.arange(grid_size, dtype=np.float32)
    gw = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(gw, gh), axis=0).reshape(2, 1, grid_size, grid_size)

    def one(d, pos):
        omega = np.arange(d // 2, dtype=np.float32)
        omega /= d / 2.0
        omega = 1.0 / 10000 ** omega
        out = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)

    return torch.from_numpy(np.concatenate([one(embed_dim // 2, grid[0]), one(embed_dim // 2, grid[1])], axis=1)).float()


def mha(q, k, v, in_w, in_b, out_w, out_b, heads):
    """nn.MultiheadAttention forward, batch-first here: q [B,Lq,E], k/v [B,Lk,E]; in_proj blocked [Q;K;V]."""
    E = q.shape[-1]
    qp = F.linear(q, in_w[:E], in_b[:E])
    kp = F.linear(k, in_w[E:2 * E], in_b[E:2 * E])
    vp = F.linear(v, in_w[2 * E:], in_b[2 * E:])
    B, Lq, _ = qp.shape
    Lk = kp.shape[1]
    hd = E // heads
    qh = qp.view(B, Lq, heads, hd).transpose(1, 2)
    kh = kp.view(B, Lk, heads, hd).transpose(1, 2)
    vh = vp.view(B, Lk, heads, hd).transpose(1, 2)
    att = torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(hd), dim=-1)
    o = (att @ vh).transpose(1, 2).reshape(B, Lq, E)
    return F.linear(o, out_w, out_b)


def resampler_forward(sd, pre, x, heads, eps):
    """qwen_visual.py:136-146 (Resampler.forward): kv_proj → ln_kv; MHA(q = ln_q(query)+pos, k = x+pos↑, v = x)."""
    pos_q = sd[pre + "pos_embed"]
    pos_k = get_abs_pos(pos_q, x.size(1))
    if pre + "kv_proj.weight" in sd:
        x = F.linear(x, sd[pre + "kv_proj.weight"])
    E = x.shape[-1]
    x = F.layer_norm(x, (E,), sd[pre + "ln_kv.weight"], sd[pre + "ln_kv.bias"], eps)
    q = F.layer_norm(sd[pre + "query"], (E,), sd[pre + "ln_q.weight"], sd[pre + "ln_q.bias"], eps)
    B = x.shape[0]
    qq = (q + pos_q).unsqueeze(0).expand(B, -1, -1)
    return mha(qq, x + pos_k.unsqueeze(0), x, sd[pre + "attn.in_proj_weight"], sd[pre + "attn.in_proj_bias"],
               sd[pre + "attn.out_proj.weight"], sd[pre + "attn.out_proj.bias"], heads)


# ---------------------------------------------------------------------------------------------------------
# Path A — VisionTransformerWithAttnPool (qwen_visual.py:325-417)
# ---------------------------------------------------------------------------------------------------------
def vit_forward(sd, cfg, x):
    """x [B,3,S,S] → [B, n_queries, output_dim]. cfg: dict(image_size, patch_size, width, layers, heads, mlp_ratio,
    n_queries, output_dim)."""
    W, heads = cfg["width"], cfg["heads"]
    hd = W // heads
    eps = 1e-6  # qwen_visual.py:358
    x = F.conv2d(x.float(), sd["conv1.weight"], stride=cfg["patch_size"])          # :393
    x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)                       # :395-396
    x = x + get_abs_pos(sd["positional_embedding"], x.size(1))                       # :398
    x = F.layer_norm(x, (W,), sd["ln_pre.weight"], sd["ln_pre.bias"], eps)           # :400
    B, L, _ = x.shape
    for i in range(cfg["layers"]):                                                    # :312-316
        p = f"transformer.resblocks.{i}."
        h = F.layer_norm(x, (W,), sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], eps)
        mixed = F.linear(h, sd[p + "attn.in_proj.weight"], sd[p + "attn.in_proj.bias"])   # :186
        mixed = mixed.view(B, L, heads, 3 * hd)                                            # :188-192 per-head interleave
        q, k, v = mixed.split(hd, dim=-1)                                                  # :195
        q = q.permute(0, 2, 1, 3) / math.sqrt(hd)                                          # :204
        k = k.permute(0, 2, 1, 3)
        v = v.permute(0, 2, 1, 3)
        att = torch.softmax(q @ k.transpose(-1, -2), dim=-1)                               # :208-209
        ctx = (att @ v).permute(0, 2, 1, 3).reshape(B, L, W)                               # :215-226
        x = x + F.linear(ctx, sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"])  # :228,280
        h = F.layer_norm(x, (W,), sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], eps)
        h = F.gelu(F.linear(h, sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"]))       # exact erf GELU :253-255
        x = x + F.linear(h, sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"])       # :281
    x = resampler_forward(sd, "attn_pool.", x, cfg["output_dim"] // 128, eps)         # :371-377,406
    x = F.layer_norm(x, (cfg["output_dim"],), sd["ln_post.weight"], sd["ln_post.bias"], eps)  # :414
    return x @ sd["proj"]                                                              # :415


# ---------------------------------------------------------------------------------------------------------
# Path B — Llama decoder (modeling_llama_xformer.py) + greedy loop + ContinuousLVLM.generate (seed_x.py)
# ---------------------------------------------------------------------------------------------------------
def rms_norm(x, w, eps):
    """transformers LlamaRMSNorm [ext] (modeling_llama_xformer.py:95): fp32 variance."""
    xf = x.float()
    return w * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps))


def rope_tables(hd, max_pos, base=10000.0):
    """modeling_llama_xformer.py:97-113."""
    inv = 1.0 / (base ** (torch.arange(0, hd, 2).float() / hd))
    fr = torch.einsum("i,j->ij", torch.arange(max_pos).float(), inv)
    emb = torch.cat((fr, fr), dim=-1)
    return emb.cos(), emb.sin()


def rotate_half(x):
    return torch.cat((-x[..., x.shape[-1] // 2:], x[..., :x.shape[-1] // 2]), dim=-1)   # :134-138


def llama_forward(sd, cfg, inputs_embeds, past=None, table_dtype=None):
    """LlamaModel.forward + lm_head (modeling_llama_xformer.py:477-609,707). inputs_embeds [1,T,H].
    past: list of (k,v) [1,nh,Tp,hd] or None. Attention: causal on prefill, unmasked for q_len == 1 (:236) — both
    are 'bottom-right aligned causal'. table_dtype: dtype the cos/sin tables are rounded to before the multiply
    (:128-131); None = fp32.
    Returns logits [1,T,V], new past, final-norm hidden states [1,T,H]."""
    H, nh, L = cfg["hidden_size"], cfg["num_attention_heads"], cfg["num_hidden_layers"]
    hd = H // nh
    eps = cfg["rms_norm_eps"]
    x = inputs_embeds.float()
    T = x.shape[1]
    Tp = 0 if past is None else past[0][0].shape[2]
    cos, sin = rope_tables(hd, Tp + T)
    if table_dtype is not None:
        cos, sin = cos.to(table_dtype).float(), sin.to(table_dtype).float()
    pos = torch.arange(Tp, Tp + T)
    cos, sin = cos[pos][None, None].to(x.device), sin[pos][None, None].to(x.device)
    new_past = []
    for i in range(L):
        p = f"model.layers.{i}."
        h = rms_norm(x, sd[p + "input_layernorm.weight"], eps)
        q = F.linear(h, sd[p + "self_attn.q_proj.weight"]).view(1, T, nh, hd).transpose(1, 2)
        k = F.linear(h, sd[p + "self_attn.k_proj.weight"]).view(1, T, nh, hd).transpose(1, 2)
        v = F.linear(h, sd[p + "self_attn.v_proj.weight"]).view(1, T, nh, hd).transpose(1, 2)
        q = q * cos + rotate_half(q) * sin                                             # :141-149
        k = k * cos + rotate_half(k) * sin
        if past is not None:
            k = torch.cat([past[i][0], k], dim=2)                                      # :215-218
            v = torch.cat([past[i][1], v], dim=2)
        new_past.append((k, v))
        s = q @ k.transpose(-1, -2) / math.sqrt(hd)
        Tk = k.shape[2]
        ii = torch.arange(T, device=x.device)[:, None] + (Tk - T)
        jj = torch.arange(Tk, device=x.device)[None, :]
        s = s.masked_fill(jj > ii, float("-inf"))
        o = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(1, T, H)
        x = x + F.linear(o, sd[p + "self_attn.o_proj.weight"])                         # :239,297
        h = rms_norm(x, sd[p + "post_attention_layernorm.weight"], eps)
        g = F.silu(F.linear(h, sd[p + "mlp.gate_proj.weight"])) * F.linear(h, sd[p + "mlp.up_proj.weight"])
        x = x + F.linear(g, sd[p + "mlp.down_proj.weight"])                            # :166-167,303
    hn = rms_norm(x, sd["model.norm.weight"], eps)                                     # :595
    logits = F.linear(hn, sd["lm_head.weight"])                                        # :707
    return logits, new_past, hn


def llama_forward_16bit_like_reference(sd, cfg, inputs_embeds, dtype):
    """What the REFERENCE computes when its scripts run the module in 16 bit (`.to(device, dtype=torch.float16)`,
    eval_img2text_seed_x_i.py:60-62,91-92), restated with the reference's dtype flow — the yardstick for "how far is 16-bit
    inference from the fp32 oracle at full depth": weights AND the residual stream in `dtype`; nn.Linear = 16-bit operands with
    fp32 accumulation, result rounded to `dtype`; LlamaRMSNorm [ext 4.30.2] = fp32 variance, x·rsqrt rounded to `dtype`, then ·weight;
    RoPE tables cast to `dtype` (:128-131) and applied in `dtype` (:141-149); attention = xformers memory_efficient_attention [ext]
    (fp32 scores / softmax / accumulation inside the kernel, output in `dtype`). Prefill only (no cache). Same shapes as llama_forward."""
    H, nh, L = cfg["hidden_size"], cfg["num_attention_heads"], cfg["num_hidden_layers"]
    hd, eps = H // nh, cfg["rms_norm_eps"]
    w = lambda k: sd[k].to(dtype)

    def norm(x, k):
        xf = x.float()
        return w(k) * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)).to(dtype)
    x = inputs_embeds.to(dtype)
    T = x.shape[1]
    cos, sin = rope_tables(hd, T)
    cos, sin = cos.to(dtype)[None, None].to(x.device), sin.to(dtype)[None, None].to(x.device)
    ii = torch.arange(T, device=x.device)
    causal = ii[None, :] > ii[:, None]
    for i in range(L):
        p = f"model.layers.{i}."
        h = norm(x, p + "input_layernorm.weight")
        q = F.linear(h, w(p + "self_attn.q_proj.weight")).view(1, T, nh, hd).transpose(1, 2)
        k = F.linear(h, w(p + "self_attn.k_proj.weight")).view(1, T, nh, hd).transpose(1, 2)
        v = F.linear(h, w(p + "self_attn.v_proj.weight")).view(1, T, nh, hd).transpose(1, 2)
        q = q * cos + rotate_half(q) * sin
        k = k * cos + rotate_half(k) * sin
        sc = (q.float() @ k.float().transpose(-1, -2)) / math.sqrt(hd)
        pr = torch.softmax(sc.masked_fill(causal, float("-inf")), dim=-1)
        o = (pr @ v.float()).to(dtype).transpose(1, 2).reshape(1, T, H)
        x = x + F.linear(o, w(p + "self_attn.o_proj.weight"))
        h = norm(x, p + "post_attention_layernorm.weight")
        g = F.silu(F.linear(h, w(p + "mlp.gate_proj.weight"))) * F.linear(h, w(p + "mlp.up_proj.weight"))
        x = x + F.linear(g, w(p + "mlp.down_proj.weight"))
    hn = norm(x, "model.norm.weight")
    return F.linear(hn, w("lm_head.weight")), hn


def logits_rule(last_id, scores, img_ids):
    """AutoImageTokenGenerationProcessor.__call__ (generation.py:19-31) for batch 1; scores [V] modified in place."""
    if last_id in img_ids[:-1]:
        out_id = img_ids[img_ids.index(last_id) + 1]
        scores[out_id] = scores.max() + 10.0
    else:
        scores[torch.tensor(img_ids[1:], dtype=torch.long)] = 0.0
    return scores


def greedy_generate(sd, cfg, input_ids, inputs_embeds, img_ids, max_new_tokens, eos_id=None, table_dtype=None,
                    force_ids=None, trace=None):
    """HF 4.30.2 greedy_search [ext] as invoked at seed_x.py:184-189 with prepare_inputs_for_generation
    (modeling_llama_xformer.py:748-779): step 0 feeds inputs_embeds, later steps the last token id.
    Returns generated ids (list) and the per-step last-layer (post-norm) hidden states [n_new, H]
    (= torch.cat of hidden_states[-1], seed_x.py:196: the state at the INPUT position of each step)."""
    emb = sd["model.embed_tokens.weight"]
    ids = list(input_ids)
    logits, past, hn = llama_forward(sd, cfg, inputs_embeds, None, table_dtype)
    hs = [hn[0]]                 # step 0 contributes T rows (sliced off at seed_x.py:197 by [input_len:])
    new = []
    for _ in range(max_new_tokens):
        scores = logits_rule(ids[-1], logits[0, -1].clone(), img_ids)
        nxt = int(torch.argmax(scores))
        if force_ids is not None:
            # teacher forcing (tests only): follow the given ids, record the oracle's own choice and the score gap
            # between its arg-max and the forced id, so 16-bit near-ties can be told apart from real mismatches
            f = int(force_ids[len(new)])
            if trace is not None:
                trace.append((nxt, f, float(scores[nxt] - scores[f]), float(scores.std())))
            nxt = f
        ids.append(nxt)
        new.append(nxt)
        if eos_id is not None and nxt == eos_id:
            break
        if len(new) == max_new_tokens:
            break
        logits, past, hn = llama_forward(sd, cfg, emb[torch.tensor([[nxt]])], past, table_dtype)
        hs.append(hn[0])
    hidden = torch.cat(hs, dim=0)
    return new, hidden


def lvlm_generate(sd_llm, sd_agent, cfg, res_cfg, input_ids, image_embeds, embeds_cmp_mask, ids_cmp_mask,
                  patch_positions, img_ids, boi_id, eoi_id, max_new_tokens, num_img_gen_tokens=64, eos_id=None,
                  table_dtype=None, force_ids=None, trace=None, return_prefill=False, tokenizer=None):
    """ContinuousLVLM.generate (seed_x.py:130-223). Pinned against the reference's own generate() executed over
    oracle/hf_generate_shim.py: tests/golden/lvlm_generate_mini.npz (tests/test_cpu_suite.py, test_oracle_vs_reference.py). input_ids: list[int]; image_embeds [n,256,4096]-like or None.
    sd_agent keys: input_resampler.*, output_resampler.*, patch_pos_embed. Returns dict like the reference plus ids."""
    emb = sd_llm["model.embed_tokens.weight"]
    x = emb[torch.tensor([input_ids])].clone()                                          # :158
    if image_embeds is not None:
        lm = resampler_forward(sd_agent, "input_resampler.", image_embeds.float(), res_cfg["in_heads"], 1e-5)  # :164
        if "patch_pos_embed" in sd_agent and patch_positions is not None:               # :165-171
            pp = patch_positions.float()
            rel = torch.mm(torch.cat([pp, 1 - pp], dim=-1) / 2, sd_agent["patch_pos_embed"]).unsqueeze(1)
            lm = lm + rel
        x[ids_cmp_mask] = lm[embeds_cmp_mask].view(-1, x.shape[-1])                     # :173
    if return_prefill:
        logits, _, hn = llama_forward(sd_llm, cfg, x, None, table_dtype)
        return {"inputs_embeds": x, "logits": logits, "hidden": hn}
    new, hidden = greedy_generate(sd_llm, cfg, input_ids, x, img_ids, max_new_tokens, eos_id, table_dtype,
                                  force_ids, trace)
    n_in = len(input_ids)
    last_hidden = hidden[n_in:]                                                          # :196-197
    gen = torch.tensor(new)
    eoi_idx = torch.where(gen == eoi_id)[0].tolist()                                     # :199
    feats = None
    if eoi_idx:
        # hidden rows are aligned with generated ids: row j = state produced when token new[j] was the INPUT.
        # last_hidden has one row fewer than `new` when generation stops on the last token (it is never fed).
        st = [last_hidden[e - num_img_gen_tokens:e] for e in eoi_idx]                    # :204-205
        feats = resampler_forward(sd_agent, "output_resampler.", torch.stack(st), res_cfg["out_heads"], 1e-5)  # :209-210
    res = {"ids": new, "has_img_output": bool(eoi_idx), "img_gen_feat": feats, "num_gen_imgs": len(eoi_idx),
           "last_hidden": last_hidden}
    if tokenizer is not None:                                                            # :201-216
        text_mask = torch.ones_like(gen, dtype=torch.bool)
        for e in eoi_idx:
            text_mask[e - num_img_gen_tokens:e] = False      # (a negative start makes this an empty slice, as in the reference)
        text_mask[gen == boi_id] = False
        res["text"] = tokenizer.decode(gen[text_mask], skip_special_tokens=False)
    return res


# ---------------------------------------------------------------------------------------------------------
# Path C head — ResamplerXLV2 (resampler.py:226-286)
# ---------------------------------------------------------------------------------------------------------
def resampler_xlv2_forward(sd, cfg, x, pre="resampler."):
    """x [B, n, embedding_dim] → (prompt_embeds [B, nq, o1+o2], pooled [B, o2]). cfg: dim, depth, dim_head, heads."""
    dim, heads, dh = cfg["dim"], cfg["heads"], cfg["dim_head"]
    B = x.shape[0]
    lat = sd[pre + "latents"].repeat(B, 1, 1)                                            # :268
    x = F.linear(x.float(), sd[pre + "proj_in.weight"], sd[pre + "proj_in.bias"])        # :273
    for i in range(cfg["depth"]):                                                         # :275-277
        a = f"{pre}layers.{i}.0."
        f = f"{pre}layers.{i}.1."
        xn = F.layer_norm(x, (dim,), sd[a + "norm1.weight"], sd[a + "norm1.bias"])      # :57-58
        ln = F.layer_norm(lat, (dim,), sd[a + "norm2.weight"], sd[a + "norm2.bias"])
        q = F.linear(ln, sd[a + "to_q.weight"])
        kv = F.linear(torch.cat((xn, ln), dim=-2), sd[a + "to_kv.weight"])              # :63-64
        k, v = kv.chunk(2, dim=-1)
        L = ln.shape[1]

        def rs(t):
            return t.view(B, t.shape[1], heads, -1).transpose(1, 2)
        q, k, v = rs(q), rs(k), rs(v)
        sc = 1 / math.sqrt(math.sqrt(dh))                                                # :68
        w = torch.softmax(((q * sc) @ (k * sc).transpose(-2, -1)).float(), dim=-1)       # :69-70
        o = (w @ v).permute(0, 2, 1, 3).reshape(B, L, -1)
        lat = F.linear(o, sd[a + "to_out.weight"]) + lat                                 # :75,276
        h = F.layer_norm(lat, (dim,), sd[f + "0.weight"], sd[f + "0.bias"])             # FeedForward :9-16
        h = F.linear(F.gelu(F.linear(h, sd[f + "1.weight"])), sd[f + "3.weight"])
        lat = h + lat                                                                    # :277
    hid = F.layer_norm(lat, (dim,), sd[pre + "norm_out.weight"], sd[pre + "norm_out.bias"])   # :279
    e1 = F.linear(hid, sd[pre + "unet_proj_1.weight"], sd[pre + "unet_proj_1.bias"])
    e2 = F.linear(hid, sd[pre + "unet_proj_2.weight"], sd[pre + "unet_proj_2.bias"])
    prompt = torch.cat([e1, e2], dim=-1)                                                  # :283
    # AttentionPool2d (:89-116): prepend mean token, add pos, MHA with separate q/k/v proj, return token 0
    p = pre + "unet_attnpool."
    t = torch.cat([hid.mean(dim=1, keepdim=True), hid], dim=1) + sd[p + "positional_embedding"][None]
    E = dim
    qh = F.linear(t[:, :1], sd[p + "q_proj.weight"], sd[p + "q_proj.bias"])
    kh = F.linear(t, sd[p + "k_proj.weight"], sd[p + "k_proj.bias"])
    vh = F.linear(t, sd[p + "v_proj.weight"], sd[p + "v_proj.bias"])
    hd = E // heads
    qh = qh.view(B, 1, heads, hd).transpose(1, 2)
    kh = kh.view(B, -1, heads, hd).transpose(1, 2)
    vh = vh.view(B, -1, heads, hd).transpose(1, 2)
    att = torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(hd), dim=-1)
    o = (att @ vh).transpose(1, 2).reshape(B, E)
    pooled = F.linear(o, sd[p + "c_proj.weight"], sd[p + "c_proj.bias"])
    return prompt, pooled
