"""(needs tools/probes/gemm_pipe4w.patch applied) Experimental tile 7 (256x256 block tile, 4 waves x 128x128 wave tiles, accumulators pinned in AGPRs, hand-ordered software
pipeline) against tile 4 (256x256, 8 waves) and 5 (256x320): correctness on edge shapes (ragged M / N, K = 64 / 128, conv,
epilogues) and speed on the UNet's C = 1280 call sites and 8192^3."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from seedx_amd import _lib, ops
from seedx_amd.llama import glu_pack_rows

lib = _lib.load()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm()).item()


# ---- correctness: tile 7 must reproduce tile 4 bit for bit (same MFMA, same k order) --------------------------------
bad = 0
for dt in (torch.bfloat16, torch.float16):
    for (M, N, K) in ((256, 256, 64), (300, 272, 128), (1000, 1040, 192), (4096, 1280, 1280), (257, 16, 64)):
        a = torch.randn(M, K, generator=g).to(dev).to(dt)
        w = (torch.randn(N, K, generator=g) * 0.1).to(dev).to(dt)
        b = torch.randn(N, generator=g).to(dev)
        r = torch.randn(M, N, generator=g).to(dev)
        for kw in (dict(), dict(bias=b, residual=r, out_dtype=torch.float32), dict(bias=b, act="gelu"),
                   dict(bias=b, act="gelu", glu=True) if N % 32 == 0 else dict()):
            outs = []
            for cfg in (4, 7):
                lib.sx_gemm_force_tile(cfg)
                outs.append(ops.gemm(a, w, **kw))
            torch.cuda.synchronize()
            ref = (a.float() @ w.float().t())
            if not torch.equal(outs[0], outs[1]):
                bad += 1
                print("MISMATCH", dt, M, N, K, sorted(kw), rel(outs[1], outs[0]))
    x = torch.randn(2, 20, 24, 64, generator=g).to(dev).to(dt)
    wc = (torch.randn(272, 9 * 64, generator=g) * 0.05).to(dev).to(dt)
    for kw in (dict(), dict(stride=2), dict(upsample=True)):
        outs = []
        for cfg in (4, 7):
            lib.sx_gemm_force_tile(cfg)
            outs.append(ops.conv3x3(x, wc, out_dtype=torch.float32, **kw))
        if not torch.equal(outs[0], outs[1]):
            bad += 1
            print("MISMATCH conv", dt, kw, rel(outs[1], outs[0]))
print("correctness: %d mismatches" % bad)


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


dt = torch.bfloat16
M, C = 32768, 1280
x = torch.randn(M, C, generator=g).to(dev).to(dt)
res = torch.randn(M, C, generator=g).to(dev)
w1 = (torch.randn(8 * C, C, generator=g) * 0.03).to(dev).to(dt)
wff1 = glu_pack_rows(w1[:4 * C].contiguous(), w1[4 * C:].contiguous())
bff1 = torch.randn(8 * C, generator=g).to(dev)
wqkv = (torch.randn(3 * C, C, generator=g) * 0.03).to(dev).to(dt)
wo = (torch.randn(C, C, generator=g) * 0.03).to(dev).to(dt)
bo = torch.randn(C, generator=g).to(dev)
gg = torch.randn(M, 4 * C, generator=g).to(dev).to(dt)
wff2 = (torch.randn(C, 4 * C, generator=g) * 0.03).to(dev).to(dt)
a8 = torch.randn(8192, 8192, generator=g).to(dev).to(dt)
b8 = (torch.randn(8192, 8192, generator=g) * 0.02).to(dev).to(dt)
xc = torch.randn(32, 32, 32, 1280, generator=g).to(dev).to(dt)
wc = (torch.randn(1280, 9 * 1280, generator=g) * 0.01).to(dev).to(dt)
cases = [("geglu N10240 K1280", lambda: ops.gemm(x, wff1, bias=bff1, act="gelu", glu=True), 2 * M * 8 * C * C),
         ("qkv N3840 K1280", lambda: ops.gemm(x, wqkv), 2 * M * 3 * C * C),
         ("out+res N1280 K1280", lambda: ops.gemm(x, wo, bias=bo, residual=res, out_dtype=torch.float32), 2 * M * C * C),
         ("ff2+res N1280 K5120", lambda: ops.gemm(gg, wff2, bias=bo, residual=res, out_dtype=torch.float32), 2 * M * 4 * C * C),
         ("8192^3", lambda: ops.gemm(a8, b8), 2 * 8192 ** 3),
         ("conv3x3 1280->1280 @32x32 B32", lambda: ops.conv3x3(xc, wc, bias=bo, out_dtype=torch.float32), 2 * M * 1280 * 9 * 1280)]
for name, fn, fl in cases:
    line = "%-30s" % name
    for cfg in (-1, 4, 7):
        lib.sx_gemm_force_tile(cfg)
        t = timeit(fn)
        line += "  %s %7.1f us %5.0f TF" % ("auto " if cfg < 0 else "tile%d" % cfg, t * 1e6, fl / t / 1e12)
    print(line)
lib.sx_gemm_force_tile(-1)
