"""Per call-site profile of the GEMM / implicit-conv family inside one config-0 bench step: every sx_gemm launch of an
eager (un-graphed) step is bracketed by HIP events on the launch stream and the launches are grouped by
(mode, M, N, K, glu, act, residual, out dtype). Prints time share, achieved TFLOP/s and the algorithmic GB/s of each group —
the table that says which shapes are MFMA-bound and which are bound by their fp32 residual / output traffic.
    python tools/gemm_shape_profile.py [--batch 16] [--unet-steps 4] [--json out.json]"""
import argparse
import collections
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--unet-steps", type=int, default=4)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    bench.BATCH = a.batch
    args = bench.parse_args(["--unet-steps", str(a.unet_steps), "--batch", str(a.batch)])
    dev = torch.device("cuda:0")
    from seedx_amd import _lib
    lib = _lib.load()
    real, real_gn, real_ln = lib.sx_gemm, lib.sx_gemm_gn, lib.sx_gemm_ln
    rec = []

    class Hook:
        def __call__(self, args_ref, *rest):       # sx_gemm(args, stream) | sx_gemm_gn(args, stats, groups, rows, fused, stream)
            g = args_ref._obj
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = {1: real, 2: real_ln}.get(len(rest), real_gn)(args_ref, *rest)      # sx_gemm_ln(args, ln, stream)
            e.record()
            n_out = g.N // 2 if g.glu else g.N
            n_st = g.n_valid if g.n_valid else n_out
            a_bytes = 2.0 * (g.B * g.Hin * g.Win * g.Cin if g.a_mode == 1 else g.M * g.K)
            byt = a_bytes + 2.0 * g.N * g.K + g.M * n_st * (4.0 if g.out_dtype == 2 else 2.0) + (4.0 * g.M * n_st if g.residual else 0.0)
            key = ("conv" if g.a_mode == 1 else "lin", g.M, g.N, g.K, int(g.glu), int(g.act), int(bool(g.residual)),
                   "f32" if g.out_dtype == 2 else "16b", int(bool(g.bias2d)), lib.sx_gemm_pick_tile(g.M, g.N, g.K, g.glu, g.a_mode))
            rec.append((key, 2.0 * g.M * g.N * g.K, byt, s, e))
            return r
    with torch.no_grad():
        w = bench.Headline(args, dev, torch.float16 if args.dtype == "fp16" else torch.bfloat16)
        w.step(0)                                                               # warm-up (graphs, caches)
        w.agent.use_graph = False
        w.adapter._loop.use_graph = False
        w.adapter._loop.chains = 1          # ONE kernel chain: with two concurrent chains an event pair also contains the other chain's kernels
        lib.sx_gemm = lib.sx_gemm_gn = lib.sx_gemm_ln = Hook()
        try:
            w.step(1)
            torch.cuda.synchronize()
        finally:
            lib.sx_gemm, lib.sx_gemm_gn, lib.sx_gemm_ln = real, real_gn, real_ln
    groups = collections.OrderedDict()
    for key, fl, byt, s, e in rec:
        g = groups.setdefault(key, [0, 0.0, 0.0, 0.0])
        g[0] += 1
        g[1] += s.elapsed_time(e) * 1e-3
        g[2] += fl
        g[3] += byt
    tot = sum(g[1] for g in groups.values())
    names = ["128x128", "128x80", "64x128", "64x64", "256x256", "256x320", "256x160", "pp256x256", "pp256x320"]
    rows = sorted(groups.items(), key=lambda kv: -kv[1][1])
    print("GEMM family: %d launches, %.3f s per step (UNet steps %d, batch %d) — %.0f TFLOP/s overall"
          % (len(rec), tot, a.unet_steps, a.batch, sum(g[2] for g in groups.values()) / tot / 1e12))
    print("%-5s %7s %6s %6s glu act res out  b2d %-8s %6s %8s %7s %8s %7s" % ("mode", "M", "N", "K", "tile", "calls", "us/call", "share", "TFLOP/s", "GB/s"))
    out = []
    for key, (n, t, fl, byt) in rows[:45]:
        mode, M, N, K, glu, act, res, od, b2d, tile = key
        print("%-5s %7d %6d %6d  %d   %d   %d  %s   %d  %-8s %6d %8.1f %6.1f%% %8.0f %7.0f"
              % (mode, M, N, K, glu, act, res, od, b2d, names[tile], n, t / n * 1e6, 100 * t / tot, fl / t / 1e12, byt / t / 1e9))
        out.append(dict(mode=mode, M=M, N=N, K=K, glu=glu, act=act, residual=res, out=od, bias2d=b2d, tile=names[tile], calls=n,
                        us_per_call=t / n * 1e6, share=t / tot, tflops=fl / t / 1e12, alg_gbps=byt / t / 1e9))
    if a.json:
        json.dump(dict(batch=a.batch, unet_steps=a.unet_steps, gemm_s_per_step=tot, rows=out), open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
