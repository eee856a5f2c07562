"""Wall time of each phase of one config-0 bench step at a given batch (synchronised between phases; GPU box).
    python tools/phase_times.py [--batch 16] [--unet-steps 50]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--unet-steps", type=int, default=50)
    a = ap.parse_args()
    bench.BATCH = a.batch
    dev, dtype, tok = torch.device("cuda:0"), torch.bfloat16, bench.BenchTokenizer()
    with torch.no_grad():
        vit, agent, adapter = bench.build_models(dev, dtype)
        inp = bench.make_inputs(dev)

        def timed(fn, n=2):
            ts = []
            for _ in range(n):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                r = fn()
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            return min(ts), r
        feats = bench.front_half(vit, agent, tok, inp, 61, dev)                   # warm-up incl. graph capture
        bench.back_half(adapter, feats, 2, 1)
        t_pre, _ = timed(lambda: bench.preprocess(inp[0], dev))
        t_a, reqs = timed(lambda: bench.requests_for(vit, inp, dev))
        t_p1, _ = timed(lambda: agent.generate_batch(tok, reqs, max_new_tokens=1, eos_token_id=None))
        t_62, _ = timed(lambda: agent.generate_batch(tok, reqs, max_new_tokens=62, eos_token_id=None))
        t_full, outs = timed(lambda: agent.generate_batch(tok, reqs, max_new_tokens=128, eos_token_id=None, force_image_at=61))
        feats = torch.cat([o["img_gen_feat"] for o in outs], dim=0)
        t_emb, _ = timed(lambda: adapter.get_image_embeds(image_embeds=feats, image_size=448))
        t_lat, lat = timed(lambda: adapter.generate(image_embeds=feats, num_inference_steps=a.unet_steps, seed=3,
                                                    output_type="latent"))
        t_img, _ = timed(lambda: adapter._finish(lat, "u8"))
        G = a.batch
        print("batch %d per step" % G)
        print("GPU preprocessing (any-res, normalise)              %9.1f ms" % (t_pre * 1e3))
        print("preprocessing + ViT (%d crops) + marker mask         %9.1f ms" % (2 * G, t_a * 1e3))
        print("input resampler + batched prefill (M=%d) + 1 token %9.1f ms" % (165 * G, t_p1 * 1e3))
        print("61 decoded tokens                                    %9.1f ms  (%.2f ms/token)" % ((t_62 - t_p1) * 1e3, (t_62 - t_p1) / 61 * 1e3))
        print("image block (65-token chunk x%d) + 2 tokens + out-resampler %9.1f ms" % (G, (t_full - t_62) * 1e3))
        print("adapter.get_image_embeds (XLV2, cached negative)     %9.1f ms" % (t_emb * 1e3))
        print("%d UNet CFG steps (batch %d)                          %9.1f ms  (%.2f ms/step)" % (a.unet_steps, 2 * G, (t_lat - t_emb) * 1e3, (t_lat - t_emb) / a.unet_steps * 1e3))
        print("VAE decode + uint8 (%d images)                       %9.1f ms" % (G, t_img * 1e3))
        print("sum                                                  %9.1f ms" % ((t_a + t_full + t_lat + t_img) * 1e3))
        # per-phase utilisation against the MI355X peaks (algorithmic work of SURVEY.md §8d ÷ phase wall time)
        PF, TB = 2500.0, 8.0
        t_dec, t_un = t_62 - t_p1, t_lat - t_emb
        rows = [("ViT-G, %d crops" % (2 * G), 2 * G * bench.FLOP_VIT_CROP / t_a / 1e12, None),
                ("prefill, %d tokens" % (165 * G), 165 * G * bench.FLOP_LLM_TOKEN / t_p1 / 1e12, None),
                ("decode, 61 steps x %d sequences" % G, 61 * G * bench.FLOP_LLM_TOKEN / t_dec / 1e12, 61 * 25.71e9 / t_dec / 1e12),
                ("UNet, %d steps x %d samples" % (a.unet_steps, 2 * G), a.unet_steps * 2 * G * bench.FLOP_UNET_SAMPLE / t_un / 1e12, None),
                ("VAE decode, %d images (fp32-product FLOPs)" % G, G * bench.FLOP_VAE_DECODE / t_img / 1e12, None)]
        print("utilisation (algorithmic work / wall time; peaks %.0f TFLOP/s dense 16-bit MFMA, %.0f TB/s HBM):" % (PF, TB))
        for name, tf, tbs in rows:
            print("  %-44s %7.1f TFLOP/s = %4.1f %% of MFMA peak%s" % (name, tf, 100 * tf / PF,
                  "" if tbs is None else "; weights streamed at %.2f TB/s = %4.1f %% of HBM peak" % (tbs, 100 * tbs / TB)))


if __name__ == "__main__":
    main()
