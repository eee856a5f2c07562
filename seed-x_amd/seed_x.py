"""ContinuousLVLM.generate on the HIP path (reference ``src/models/mllm/seed_x.py:22-46,130-234``).

Same constructor / ``from_pretrained`` / ``generate`` signature and return dict as the reference. The greedy loop
of HF ``GenerationMixin.generate`` [ext, transformers 4.30.2] plus ``AutoImageTokenGenerationProcessor``
(generation.py:9-31) is restated as a device-resident loop: one hipGraph replay per token, a single 4-byte
read-back per token for the EOS / ``<img>`` test (the reference performs ≥45 device→host syncs per token).
When ``<img>`` is emitted the 64 forced ``<img_i>`` tokens are run as ONE 65-token causal chunk on the MFMA path
(identical math, 64× the arithmetic intensity).
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
    @torch.no_grad()
    def generate(self, tokenizer, prompt=None, input_ids=None, image_embeds=None, embeds_cmp_mask=None,
                 ids_cmp_mask=None, logits_processor=None, num_img_gen_tokens=64, temperature=0.7, num_beams=1,
                 max_new_tokens=120, top_p=0.5, dtype=torch.float16, device='cuda', patch_positions=None,
                 eos_token_id="auto", force_image_at=None):
        """Greedy (do_sample=False, num_beams=1 — temperature/top_p are inert in the reference too, seed_x.py:175-189).
        ``eos_token_id``: "auto" → tokenizer.eos_token_id; None disables the EOS stop (fixed-length benchmarking)."""
        assert logits_processor is None, "the AutoImageTokenGenerationProcessor rule is fused on the device"
        assert num_beams == 1
        llm = self.llm
        dev = llm.device
        if prompt is not None:
            input_ids = tokenizer(prompt, return_tensors="pt").input_ids
        if isinstance(input_ids, torch.Tensor):
            input_ids = input_ids.reshape(-1).tolist()
        else:
            input_ids = list(input_ids[0]) if len(input_ids) and isinstance(input_ids[0], (list, tuple)) else list(input_ids)
        T = len(input_ids)
        H = llm.H
        ids_dev = torch.tensor(input_ids, dtype=torch.int32, device=dev)
        x = ops.embedding(ids_dev, llm._pack()["embed"])                                    # [T, H] fp32 (:158)

        if image_embeds is not None:
            assert embeds_cmp_mask is not None and ids_cmp_mask is not None
            lm = self.input_resampler(image_embeds.to(dev))                                 # [n, nq, H] fp32 (:164)
            n, nq, _ = lm.shape
            if self.add_patch_pos:                                                          # :165-171
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

        img_ids = tokenizer.encode(''.join([BOI_TOKEN] + [IMG_TOKEN.format(i) for i in range(num_img_gen_tokens)]
                                           + [EOI_TOKEN]), add_special_tokens=False)        # generation.py:15-17
        boi_id, eoi_id = img_ids[0], img_ids[-1]
        if eos_token_id == "auto":
            eos_token_id = getattr(tokenizer, "eos_token_id", None)
        img_ids_dev = torch.tensor(img_ids, dtype=torch.int32, device=dev)
        out_ids = torch.full((max_new_tokens + 2,), -1, dtype=torch.int32, device=dev)
        hid = torch.zeros((max_new_tokens + 2, H), dtype=torch.float32, device=dev)        # row k = state at input new[k-1]

        # ---- prefill + first token --------------------------------------------------------------------------
        llm.reset()
        P = llm._P
        logits, _ = llm.forward_embeds(x)
        P["cur"].fill_(input_ids[-1])
        ops.greedy_next(logits, llm.V, img_ids_dev, P["cur"], P["cur"], out_ids, P["step"])
        ops.add_i32(P["step"], 1)
        n_new = 1
        cur = int(P["cur"].item())

        def maybe_force(cur):
            # synthetic-weights benchmarking only: random-init weights never emit <img>, so the transcript is pinned by
            # overwriting generated token #force_image_at with <img> AFTER its full forward/lm_head/argmax has run
            # (no work is skipped). Never set with real checkpoints.
            if force_image_at is not None and n_new - 1 == force_image_at:
                P["cur"].fill_(boi_id)
                out_ids[n_new - 1] = boi_id
                return boi_id
            return cur
        cur = maybe_force(cur)
        # ---- token loop -------------------------------------------------------------------------------------------
        while n_new < max_new_tokens and not (eos_token_id is not None and cur == eos_token_id):
            nchunk = num_img_gen_tokens + 1
            if self.chunk_forced_image_tokens and cur == boi_id and n_new + nchunk <= max_new_tokens:
                # inputs [<img>, <img_0> … <img_63>] as one causal chunk; outputs are forced (generation.py:23-26)
                chunk = torch.tensor([boi_id] + img_ids[1:-1], dtype=torch.int32, device=dev)
                xe = ops.embedding(chunk, P["embed"])
                _, hn = llm.forward_embeds(xe, need_logits=False)
                hid[n_new:n_new + nchunk] = hn                                               # plumbing copy
                out_ids[n_new:n_new + nchunk] = torch.tensor(img_ids[1:], dtype=torch.int32, device=dev)
                n_new += nchunk
                P["step"].fill_(n_new)
                P["cur"].fill_(eoi_id)
                cur = eoi_id
                continue
            llm.decode_step(img_ids_dev, out_ids, hid, use_graph=self.use_graph)
            n_new += 1
            cur = maybe_force(int(P["cur"].item()))

        generate_ids = out_ids[:n_new].cpu().long()
        last_hidden = hid[1:n_new]                                                           # seed_x.py:196-197
        eoi_indices = torch.where(generate_ids == eoi_id)[0].tolist()                        # :199
        num_gen_imgs = len(eoi_indices)
        text_mask = torch.ones_like(generate_ids, dtype=torch.bool)
        has_img_output = num_gen_imgs > 0
        img_gen_feat = None
        if has_img_output:
            feats = []
            for e in eoi_indices:
                feats.append(last_hidden[e - num_img_gen_tokens:e])                          # :204
                text_mask[e - num_img_gen_tokens:e] = False
            img_gen_feat = self.output_resampler(torch.stack(feats))                         # :209-210
            img_gen_feat = ops.cast(img_gen_feat.contiguous(), self.dtype)
        text_mask[generate_ids == boi_id] = False
        text_ids = generate_ids[text_mask]
        text = tokenizer.decode(text_ids, skip_special_tokens=False)                         # :214-216
        return {'text': text, 'has_img_output': has_img_output, 'img_gen_feat': img_gen_feat,
                'num_gen_imgs': num_gen_imgs, 'generate_ids': generate_ids, 'last_hidden_states': last_hidden}
