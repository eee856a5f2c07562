"""N eager (un-graphed, ONE kernel chain) UNet CFG steps at a given batch — the target of the rocprofv3 passes (--kernel-trace
--stats, and the FETCH_SIZE / WRITE_SIZE --pmc passes of tools/pmc_unet.sh).

--alg-json FILE: additionally hook sx_gemm / sx_attention (the same byte / FLOP formulas as bench.py's instrumented pass) and
write the ALGORITHMIC bytes and FLOPs of exactly the launches this command issues, per kernel family — so that the PMC traffic
of these launches is compared with the algorithmic bytes of THESE launches and not of a different population."""
import argparse
import json
import os
import sys

ROOT = os.environ.get("SX_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
ap.add_argument("--alg-json", default=None)
a = ap.parse_args()
import bench
bench.BATCH = a.batch
bench.USE_VAE = False
dt = torch.float16 if a.dtype == "fp16" else torch.bfloat16
dev = torch.device("cuda:0")
with torch.no_grad():
    _, _, adapter = bench.build_models(dev, dt, need=("adapter",))
    adapter.use_graph = False
    adapter._loop.use_graph = False
    adapter._loop.chains = 1
    per_launch = []       # one record per sx_gemm launch, in launch order: [mode, M, N, K, glu, residual, fp32 out, read bytes, written bytes]
    fam = {"gemm": {"launches": 0, "bytes": 0.0, "flop": 0.0}, "attention": {"launches": 0, "bytes": 0.0, "flop": 0.0}}
    if a.alg_json:
        from seedx_amd import _lib
        lib = _lib.load()
        real_gemm, real_gemm_gn, real_gemm_ln, real_attn = lib.sx_gemm, lib.sx_gemm_gn, lib.sx_gemm_ln, lib.sx_attention

        def h_gemm(args_ref, *rest):
            g = args_ref._obj
            n_out = g.N // 2 if g.glu else g.N
            n_st = g.n_valid if g.n_valid else n_out
            a_bytes = 2.0 * (g.B * g.Hin * g.Win * g.Cin if g.a_mode == 1 else g.M * g.K)          # operands once + output once
            byt = a_bytes + 2.0 * g.N * g.K + g.M * n_st * (4.0 if g.out_dtype == 2 else 2.0) + (4.0 * g.M * n_st if g.residual else 0.0)
            wr = g.M * n_st * (4.0 if g.out_dtype == 2 else 2.0)
            if len(rest) == 2 and rest[0]._obj.row_stats_out:      # sx_gemm_ln producer: the 16-bit copy of the output is written too
                wr += 2.0 * g.M * n_st
                byt += 2.0 * g.M * n_st
            f = fam["gemm"]
            f["launches"] += 1; f["bytes"] += byt; f["flop"] += 2.0 * g.M * g.N * g.K
            per_launch.append(["conv" if g.a_mode else "lin", g.M, g.N, g.K, int(g.glu), int(bool(g.residual)), int(g.out_dtype == 2),
                               byt - wr, wr])
            return {1: real_gemm, 2: real_gemm_ln}.get(len(rest), real_gemm_gn)(args_ref, *rest)

        def h_attn(args_ref, stream):
            g = args_ref._obj
            f = fam["attention"]
            f["launches"] += 1
            f["bytes"] += 2.0 * g.B * g.H * g.D * (2 * g.Sq + 2 * g.Skv)                           # Q + O, K + V once
            f["flop"] += 4.0 * g.B * g.H * g.Sq * g.Skv * g.D
            return real_attn(args_ref, stream)
        lib.sx_gemm, lib.sx_gemm_gn, lib.sx_gemm_ln, lib.sx_attention = h_gemm, h_gemm, h_gemm, h_attn
    feats = torch.randn(a.batch, 64, 4096, device=dev).to(dt)
    adapter.generate(image_embeds=feats, num_inference_steps=a.steps, seed=1, output_type="latent")
    torch.cuda.synchronize()
    if a.alg_json:
        lib.sx_gemm, lib.sx_gemm_gn, lib.sx_gemm_ln, lib.sx_attention = real_gemm, real_gemm_gn, real_gemm_ln, real_attn
        json.dump({"batch": a.batch, "steps": a.steps, "dtype": a.dtype, "families": fam, "gemm_launches": per_launch}, open(a.alg_json, "w"))
