"""ContinuousLVLM.generate on the HIP path (reference ``src/models/mllm/seed_x.py:22-46,130-234``).

Same constructor / ``from_pretrained`` / ``generate`` signature and return dict as the reference. The greedy loop
of HF ``GenerationMixin.generate`` [ext, transformers 4.30.2] plus ``AutoImageTokenGenerationProcessor``
(generation.py:9-31) is restated as a device-resident loop: one hipGraph replay per token, a single small
read-back per token for the EOS / ``<img>`` test (the reference performs ≥45 device→host syncs per token).
When ``<img>`` is emitted the 64 forced ``<img_i>`` tokens are run as ONE 65-token causal chunk on the MFMA path
(identical math, 64× the arithmetic intensity).
``generate_batch`` runs G independent requests in lock step (one weight stream from HBM per token step for all G).
"""
import torch

from . import ops

BOI_TOKEN = '<img>'
EOI_TOKEN = '</img>'
IMG_TOKEN = '<img_{:05d}>'


class ContinuousLVLM:
    def __init__(self, llm, input_resampler, output_resampler, lm_loss_scale=1.0, rec_loss_scale=1.0,
                 add_patch_pos=False, vit_down=False, mse=False):
        self.llm = llm
        self.input_resampler = input_resampler
        self.output_resampler = output_resampler
        self.add_patch_pos = add_patch_pos
        self.vit_down = vit_down
        self.patch_pos_embed = None      # host fp32 [4, dim] (seed_x.py:43-45)
        self.device, self.dtype = None, torch.float16
        self.use_graph = True
        self.chunk_forced_image_tokens = True
        self._conv = {}                  # sequence → (token ids, row fingerprints, LLM cache epoch) held by its KV cache
        self._sig_w = {}
        self.last_prefill_tokens = []

    @classmethod
    def from_pretrained(cls, llm, input_resampler, output_resampler, pretrained_model_path=None, **kwargs):
        model = cls(llm=llm, input_resampler=input_resampler, output_resampler=output_resampler, **kwargs)
        if pretrained_model_path is not None:
            ckpt = torch.load(pretrained_model_path, map_location="cpu")   # agent/pytorch_model.bin (:231-233)
            model.load_state_dict(ckpt)
        return model

    def load_state_dict(self, sd, strict=True):
        """Keys: input_resampler.*, output_resampler.*, patch_pos_embed (+ optional llm.* which is ignored here: the
        released checkpoints ship the LLM as a merged HF directory, llm_seed_x_i.yaml)."""
        self.input_resampler.load_state_dict(sd, prefix="input_resampler.", strict=strict)
        self.output_resampler.load_state_dict(sd, prefix="output_resampler.", strict=strict)
        if self.add_patch_pos:
            if "patch_pos_embed" not in sd:
                raise KeyError("ContinuousLVLM: missing key patch_pos_embed")
            self.patch_pos_embed = sd["patch_pos_embed"].detach().float().cpu()
        return [], []

    def to(self, device=None, dtype=None):
        if device is not None:
            self.device = torch.device(device)
        if dtype is not None:
            self.dtype = dtype
        for m in (self.llm, self.input_resampler, self.output_resampler):
            m.to(self.device, self.dtype)
        return self

    def eval(self):
        return self

    # ------------------------------------------------------------------------------------------------------------
    def _prompt_embeds(self, tokenizer, req):
        """embed_tokens + input resampler + patch-position term + scatter (seed_x.py:154-173) for ONE request."""
        llm = self.llm
        dev, H = llm.device, llm.H
        input_ids = req.get("input_ids")
        if req.get("prompt") is not None:
            input_ids = tokenizer(req["prompt"], return_tensors="pt").input_ids
        if isinstance(input_ids, torch.Tensor):
            input_ids = input_ids.reshape(-1).tolist()
        else:
            input_ids = list(input_ids[0]) if len(input_ids) and isinstance(input_ids[0], (list, tuple)) else list(input_ids)
        ids_dev = torch.tensor(input_ids, dtype=torch.int32, device=dev)
        x = ops.embedding(ids_dev, llm._pack()["embed"])                                    # [T, H] fp32 (:158)
        image_embeds = req.get("image_embeds")
        if image_embeds is not None:
            embeds_cmp_mask, ids_cmp_mask = req.get("embeds_cmp_mask"), req.get("ids_cmp_mask")
            assert embeds_cmp_mask is not None and ids_cmp_mask is not None
            lm = self.input_resampler(image_embeds.to(dev))                                 # [n, nq, H] fp32 (:164)
            n, nq, _ = lm.shape
            if self.add_patch_pos:                                                          # :165-171
                patch_positions = req.get("patch_positions")
                assert patch_positions is not None
                pp = patch_positions.detach().float().cpu()
                rel = torch.mm(torch.cat([pp, 1 - pp], dim=-1) / 2, self.patch_pos_embed)   # host glue, [n, H]
                rel = rel.to(dev).unsqueeze(1).expand(n, nq, H).contiguous()
                lm = ops.add(lm.contiguous(), rel)
            sel = torch.nonzero(embeds_cmp_mask.detach().cpu().reshape(-1)).reshape(-1).tolist()
            rows = torch.nonzero(ids_cmp_mask.detach().cpu().reshape(-1)).reshape(-1).to(torch.int32)
            assert rows.numel() == len(sel) * nq, "ids_cmp_mask / embeds_cmp_mask mismatch"
            src = lm if len(sel) == n else lm[torch.tensor(sel, device=dev)]
            ops.scatter_rows(src.reshape(-1, H).contiguous(), rows.to(dev), x)              # :173
        return input_ids, x

    @torch.no_grad()
    def _row_sig(self, x):
        """Per-row fingerprint of prompt embeddings (fp32 [T, H]) used to validate a cached prefix: two independent
        position-weighted sums of the rows' bit patterns (int64 [T, 2]) — a plain sum would not see permuted columns."""
        bits = x.view(torch.int32).to(torch.int64)
        w = self._sig_w.get((x.shape[1], x.device))
        if w is None:
            gen = torch.Generator().manual_seed(0x5EED)
            w = (torch.randint(1, 1 << 20, (x.shape[1], 2), generator=gen, dtype=torch.int64) * 2 + 1).to(x.device)
            self._sig_w[(x.shape[1], x.device)] = w
        return torch.stack([(bits * w[:, 0]).sum(dim=1), (bits * w[:, 1]).sum(dim=1)], dim=1)

    def _reuse_prefix(self, prompts):
        """Cross-turn KV reuse (no reference counterpart: seed_x.py:184-189 re-prefills the whole conversation every turn).
        For sequence g the cache still holds the previous call's prompt + the generated tokens that were fed back. The new
        prompt's longest prefix whose token ids AND embedding rows (image features included) are identical to what produced
        those cache entries is kept; only the rest is prefilled. Returns the prefix length per sequence."""
        starts = []
        for g, (ids, x) in enumerate(prompts):
            prev = self._conv.get(g)
            p = 0
            if prev is not None and prev[2] != self.llm.kv_epoch:
                prev = None                      # the cache was reset / written behind our back (llm.reset, llm.forward, ...)
            if prev is not None:
                old_ids, old_sig, _ = prev
                m = min(len(old_ids), len(ids) - 1)                                   # >= 1 token must be forwarded
                while p < m and old_ids[p] == ids[p]:
                    p += 1
                if p:
                    same = (self._row_sig(x[:p]) == old_sig[:p]).all(dim=1).to(torch.int32)
                    p = int(torch.cumprod(same, 0).sum().item())
            starts.append(p)
        return starts

    def _remember(self, g, prompt, gen_ids_fed):
        """Records what sequence g's KV cache now holds: the prompt rows and the generated tokens that were fed back."""
        ids, x = prompt
        sig = self._row_sig(x)
        if len(gen_ids_fed):
            P = self.llm._P
            e = ops.embedding(torch.tensor(gen_ids_fed, dtype=torch.int32, device=x.device), P["embed"])
            sig = torch.cat([sig, self._row_sig(e)])
        self._conv[g] = (list(ids) + list(gen_ids_fed), sig, self.llm.kv_epoch)

    @torch.no_grad()
    def generate_batch(self, tokenizer, requests, num_img_gen_tokens=64, max_new_tokens=120, eos_token_id="auto",
                       force_image_at=None, reuse_cache=False):
        """G = len(requests) = llm.G independent requests decoded in lock step. Each request is a dict with the
        ``generate`` keyword arguments (input_ids | prompt, image_embeds, embeds_cmp_mask, ids_cmp_mask,
        patch_positions). Returns one reference-style result dict per request.
        ``reuse_cache``: keep each sequence's KV cache across calls and prefill only the part of the new prompt that is not
        already in it (multi-turn conversations; results are identical to a full re-prefill)."""
        llm = self.llm
        dev, H, G = llm.device, llm.H, len(requests)
        P = llm._pack()
        assert G == llm.G, f"the LLM was built for max_batch={llm.G} lock-step sequences, got {G} requests"
        img_ids = tokenizer.encode(''.join([BOI_TOKEN] + [IMG_TOKEN.format(i) for i in range(num_img_gen_tokens)]
                                           + [EOI_TOKEN]), add_special_tokens=False)        # generation.py:15-17
        boi_id, eoi_id = img_ids[0], img_ids[-1]
        if eos_token_id == "auto":
            eos_token_id = getattr(tokenizer, "eos_token_id", None)
        img_ids_dev = torch.tensor(img_ids, dtype=torch.int32, device=dev)
        nchunk = num_img_gen_tokens + 1
        rows = 2 * max_new_tokens + nchunk + 8          # finished sequences keep stepping until the slowest one ends
        out_ids = torch.full((G, rows), -1, dtype=torch.int32, device=dev)
        hid = torch.zeros((G, rows, H), dtype=torch.float32, device=dev)                    # row k = state at input new[k-1]

        # ---- prefill every request (ONE batched pass: M = sum of the prompt lengths), first token --------------------
        prompts = [self._prompt_embeds(tokenizer, req) for req in requests]
        starts = self._reuse_prefix(prompts) if reuse_cache else [0] * G
        if not reuse_cache:
            llm.reset()
            self._conv = {}
        xs, last_ids = [], []
        for g, (input_ids, x) in enumerate(prompts):
            assert len(input_ids) + rows <= llm.Tmax, "KV cache too small for prompt + max_new_tokens"
            if reuse_cache:
                llm.set_position(g, starts[g])
            xs.append(x[starts[g]:])
            last_ids.append(input_ids[-1])
        P["step"].zero_()
        logits, _ = llm.forward_embeds_batch(xs, list(range(G)))
        logits = logits.contiguous()
        self.last_prefill_tokens = [int(x.shape[0]) for x in xs]
        P["cur"].copy_(torch.tensor(last_ids, dtype=torch.int32))
        ops.greedy_next_b(logits, llm.V, img_ids_dev, P["cur"], out_ids, P["step"])
        ops.add_i32(P["step"], 1)
        n_new = [1] * G
        cur = P["cur"].tolist()
        llm.comm.check()

        def force(g):
            # synthetic-weights benchmarking only: random-init weights never emit <img>, so the transcript is pinned by
            # overwriting generated token #force_image_at with <img> AFTER its full forward/lm_head/argmax has run
            # (no work is skipped). Never set with real checkpoints.
            if force_image_at is not None and n_new[g] - 1 == force_image_at:
                P["cur"][g] = boi_id
                out_ids[g, n_new[g] - 1] = boi_id
                cur[g] = boi_id

        def finished(g):
            return n_new[g] >= max_new_tokens or (eos_token_id is not None and cur[g] == eos_token_id)
        for g in range(G):
            force(g)
        done = [finished(g) for g in range(G)]
        final_n = [n_new[g] if done[g] else None for g in range(G)]
        # ---- token loop ---------------------------------------------------------------------------------------------
        while not all(done):
            hit = [g for g in range(G) if not done[g] and self.chunk_forced_image_tokens and cur[g] == boi_id
                   and n_new[g] + nchunk <= max_new_tokens]
            if hit:
                # inputs [<img>, <img_0> … <img_63>] as one causal chunk per sequence, all such sequences in ONE pass;
                # the outputs are forced (generation.py:23-26)
                chunk = torch.tensor([boi_id] + img_ids[1:-1], dtype=torch.int32, device=dev)
                xe = ops.embedding(chunk, P["embed"])
                _, hns = llm.forward_embeds_batch([xe] * len(hit), hit, need_logits=False)
                forced = torch.tensor(img_ids[1:], dtype=torch.int32, device=dev)
                for g, hn in zip(hit, hns):
                    hid[g, n_new[g]:n_new[g] + nchunk] = hn                                  # plumbing copy
                    out_ids[g, n_new[g]:n_new[g] + nchunk] = forced
                    n_new[g] += nchunk
                    P["step"][g] = n_new[g]
                    P["cur"][g] = eoi_id
                    cur[g] = eoi_id
                    if finished(g):
                        done[g], final_n[g] = True, n_new[g]
            if all(done):
                break
            llm.decode_step(img_ids_dev, out_ids, hid, use_graph=self.use_graph)            # one token for every sequence
            cur = P["cur"].tolist()                                                          # the only read-back per step
            llm.comm.check()                              # (tensor-parallel only: + 4 bytes) a timed-out collective must not pass
            for g in range(G):
                if done[g]:
                    continue
                n_new[g] += 1
                force(g)
                if finished(g):
                    done[g], final_n[g] = True, n_new[g]

        results = []
        for g in range(G):
            n = final_n[g]
            generate_ids = out_ids[g, :n].cpu().long()
            if reuse_cache:
                self._remember(g, prompts[g], generate_ids[:n - 1].tolist())          # the last new token was never fed
            last_hidden = hid[g, 1:n]                                                        # seed_x.py:196-197
            eoi_indices = torch.where(generate_ids == eoi_id)[0].tolist()                    # :199
            text_mask = torch.ones_like(generate_ids, dtype=torch.bool)
            img_gen_feat = None
            if eoi_indices:
                feats = []
                for e in eoi_indices:
                    feats.append(last_hidden[e - num_img_gen_tokens:e])                      # :204
                    text_mask[e - num_img_gen_tokens:e] = False
                img_gen_feat = self.output_resampler(torch.stack(feats))                     # :209-210
                img_gen_feat = ops.cast(img_gen_feat.contiguous(), self.dtype)
            text_mask[generate_ids == boi_id] = False
            text = tokenizer.decode(generate_ids[text_mask], skip_special_tokens=False)      # :214-216
            results.append({'text': text, 'has_img_output': len(eoi_indices) > 0, 'img_gen_feat': img_gen_feat,
                            'num_gen_imgs': len(eoi_indices), 'generate_ids': generate_ids,
                            'last_hidden_states': last_hidden})
        return results

    @torch.no_grad()
    def generate(self, tokenizer, prompt=None, input_ids=None, image_embeds=None, embeds_cmp_mask=None,
                 ids_cmp_mask=None, logits_processor=None, num_img_gen_tokens=64, temperature=0.7, num_beams=1,
                 max_new_tokens=120, top_p=0.5, dtype=torch.float16, device='cuda', patch_positions=None,
                 eos_token_id="auto", force_image_at=None, reuse_cache=False):
        """Reference signature (seed_x.py:130-145). Greedy (do_sample=False, num_beams=1 — temperature/top_p are inert in
        the reference too, :175-189). ``eos_token_id``: "auto" → tokenizer.eos_token_id; None disables the EOS stop."""
        assert logits_processor is None, "the AutoImageTokenGenerationProcessor rule is fused on the device"
        assert num_beams == 1
        assert self.llm.G == 1, "this LLM was built for lock-step batches: use generate_batch()"
        req = dict(prompt=prompt, input_ids=input_ids, image_embeds=image_embeds, embeds_cmp_mask=embeds_cmp_mask,
                   ids_cmp_mask=ids_cmp_mask, patch_positions=patch_positions)
        return self.generate_batch(tokenizer, [req], num_img_gen_tokens, max_new_tokens, eos_token_id, force_image_at,
                                   reuse_cache)[0]
