"""Where a tile's time goes inside the GEMM kernel: per-workgroup s_memtime stamps (start, first k-tile landed, main loop
done, stores issued) through the sx_gemm_debug_stamps hook, for the UNet's K = 1280 call sites at CFG batch 32.
Also reports how synchronised the workgroups of one launch are (spread of start / end times per round)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seedx_amd import _lib, ops
from seedx_amd.llama import glu_pack_rows

dev, dt = torch.device("cuda:0"), torch.bfloat16
lib = _lib.load()
M, Cc = 32768, 1280
x = torch.randn(M, Cc, device=dev).to(dt)
res = torch.randn(M, Cc, device=dev)
wo = (torch.randn(Cc, Cc, device=dev) * 0.03).to(dt)
bo = torch.randn(Cc, device=dev)
wqkv = (torch.randn(3 * Cc, Cc, device=dev) * 0.03).to(dt)
w1 = (torch.randn(8 * Cc, Cc, device=dev) * 0.03).to(dt)
wff1 = glu_pack_rows(w1[:4 * Cc].contiguous(), w1[4 * Cc:].contiguous())
bff1 = torch.randn(8 * Cc, device=dev)
g = torch.randn(M, 4 * Cc, device=dev).to(dt)
wff2 = (torch.randn(Cc, 4 * Cc, device=dev) * 0.03).to(dt)
cases = [("geglu 256x256", lambda: ops.gemm(x, wff1, bias=bff1, act="gelu", glu=True)),
         ("qkv 256x320", lambda: ops.gemm(x, wqkv)),
         ("out+res 256x320", lambda: ops.gemm(x, wo, bias=bo, residual=res, out_dtype=torch.float32)),
         ("ff2+res K5120", lambda: ops.gemm(g, wff2, bias=bo, residual=res, out_dtype=torch.float32))]
buf = torch.zeros(65536 * 4, dtype=torch.int64, device=dev)
for name, fn in cases:
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    buf.zero_()
    lib.sx_gemm_debug_stamps(C.c_void_p(buf.data_ptr()))
    fn()
    torch.cuda.synchronize()
    lib.sx_gemm_debug_stamps(None)
    t = buf.view(-1, 4).cpu()
    t = t[t[:, 3] > 0].double()
    n = t.shape[0]
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); fn(); e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3
    # s_memtime counters of different XCDs are not aligned: only differences inside one workgroup are meaningful. Every CU
    # runs n/256 tiles back to back, so launch time ≈ (n/256) x mean tile time fixes the tick length.
    total = (t[:, 3] - t[:, 0])
    tick = us / (n / 256.0 * total.mean().item())
    pro, main, epi = (t[:, 1] - t[:, 0]) * tick, (t[:, 2] - t[:, 1]) * tick, (t[:, 3] - t[:, 2]) * tick
    print("%-16s %5d tiles, launch %7.1f us (tick %.5f us = %.0f MHz) | per tile: start→first k-tile landed %5.2f us, main loop "
          "%6.2f us, epilogue %5.2f us | tile total %6.2f us (min %.2f max %.2f)"
          % (name, n, us, tick, 1.0 / tick, pro.mean(), main.mean(), epi.mean(), (pro + main + epi).mean(),
             (total * tick).min(), (total * tick).max()))
