"""Batched-decode skinny GEMM in situ: the four weight matrices of a Llama-13B layer, `LAYERS` distinct layers streamed one
after the other (2.6 GB: nothing is cache-resident, unlike tools/bench_gemv.py), 16 rows — row-major weights against the
decode-tile layout (ops.pack_decode_tiles). Prints µs and TB/s per matrix and checks the two layouts give identical results."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seedx_amd import ops
from seedx_amd.llama import glu_pack_rows

dev, dt = torch.device("cuda:0"), torch.bfloat16
LAYERS, M, H, I = 4, 16, 5120, 13824
g = torch.Generator().manual_seed(0)
shapes = [("qkv", 3 * H, H, False), ("o", H, H, False), ("gate|up", 2 * I, H, True), ("down", H, I, False)]
W, T = [], []
for _ in range(LAYERS):
    lw, lt = [], []
    for name, N, K, glu in shapes:
        w = (torch.randn(N, K, device=dev) * 0.02).to(dt)
        if glu:
            w = glu_pack_rows(w[:N // 2].contiguous(), w[N // 2:].contiguous())
        lw.append(w)
        lt.append(ops.pack_decode_tiles(w))
    W.append(lw)
    T.append(lt)
xs = {K: torch.randn(M, K, device=dev).to(dt) for K in (H, I)}
ok = True
for i, (name, N, K, glu) in enumerate(shapes):
    a = ops.gemv(xs[K], W[0][i], act="silu" if glu else None, glu=glu)
    b = ops.gemv(xs[K], W[0][i], act="silu" if glu else None, glu=glu, w_tiles=T[0][i])
    ok &= torch.equal(a, b)
print("decode-tile layout == row-major results:", ok)
for label, use_tiles in (("row-major", False), ("decode tiles", True), ("row-major", False), ("decode tiles", True)):
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in shapes] for _ in range(LAYERS)]
    for rep in range(3):                                    # last repetition is the one read
        for l in range(LAYERS):
            for i, (name, N, K, glu) in enumerate(shapes):
                s, e = ev[l][i]
                s.record()
                ops.gemv(xs[K], W[l][i], act="silu" if glu else None, glu=glu, w_tiles=T[l][i] if use_tiles else None)
                e.record()
    torch.cuda.synchronize()
    line, tot_us, tot_b = "%-13s" % label, 0.0, 0.0
    for i, (name, N, K, glu) in enumerate(shapes):
        us = sum(ev[l][i][0].elapsed_time(ev[l][i][1]) for l in range(LAYERS)) / LAYERS * 1e3
        line += "  %s %6.1f us %4.2f TB/s" % (name, us, N * K * 2 / us / 1e6)
        tot_us += us
        tot_b += N * K * 2
    print(line + "  | layer %6.1f us %4.2f TB/s" % (tot_us, tot_b / tot_us / 1e6))
