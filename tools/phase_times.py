"""Wall time of each phase of one bench generation (synchronised between phases; run on the GPU box)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench


def main():
    dev = torch.device("cuda:0")
    dtype = torch.bfloat16
    tok = bench.BenchTokenizer()
    with torch.no_grad():
        vit, agent, adapter = bench.build_models(dev, dtype)
        inp = bench.make_inputs(dev)
        for _ in range(2):
            bench.one_generation(vit, agent, adapter, tok, inp, 50, 61, seed=1)
        torch.cuda.synchronize()
        image, patch_pos, ids, mask = inp

        def timed(fn, n=3):
            ts = []
            for _ in range(n):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                r = fn()
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            return min(ts), r
        t_vit, emb = timed(lambda: vit(image))
        kw = dict(input_ids=[ids], image_embeds=emb, embeds_cmp_mask=torch.tensor([True, True]), ids_cmp_mask=mask,
                  patch_positions=patch_pos, eos_token_id=None)
        t_pre, _ = timed(lambda: agent.generate(tok, max_new_tokens=1, **kw))
        t_62, _ = timed(lambda: agent.generate(tok, max_new_tokens=62, **kw))
        t_full, out = timed(lambda: agent.generate(tok, max_new_tokens=128, force_image_at=61, **kw))
        agent.chunk_forced_image_tokens = False
        t_nochunk, _ = timed(lambda: agent.generate(tok, max_new_tokens=128, force_image_at=61, **kw), n=1)
        agent.chunk_forced_image_tokens = True
        feats = out["img_gen_feat"]
        t_emb, _ = timed(lambda: adapter.get_image_embeds(image_embeds=feats, image_size=448))
        t_unet, _ = timed(lambda: adapter.generate(image_embeds=feats, num_inference_steps=50, seed=3), n=2)
        print("ViT (2 crops)                       %8.1f ms" % (t_vit * 1e3))
        print("LLM input resampler + prefill(165)  %8.1f ms" % (t_pre * 1e3))
        print("LLM 61 decoded tokens               %8.1f ms  (%.2f ms/token)" % ((t_62 - t_pre) * 1e3, (t_62 - t_pre) / 61 * 1e3))
        print("LLM image block (65-token chunk) + 2 tokens + output resampler %8.1f ms" % ((t_full - t_62) * 1e3))
        print("   same without chunking (66 single-token steps)             %8.1f ms" % ((t_nochunk - t_62) * 1e3))
        print("adapter.get_image_embeds (XLV2, cached negative)             %8.1f ms" % (t_emb * 1e3))
        print("adapter.generate 50 UNet steps      %8.1f ms  (%.2f ms/step)" % (t_unet * 1e3, (t_unet - t_emb) / 50 * 1e3))
        print("sum                                 %8.1f ms" % ((t_vit + t_full + t_unet) * 1e3))


if __name__ == "__main__":
    main()
