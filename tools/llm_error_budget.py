"""Per-site error budget of the 40-layer Llama-13B-dim decoder in 16-bit arithmetic, on the CPU.

Emulates the HIP path's dtype flow (fp32 residual stream / norms / softmax / accumulation, 16-bit MFMA operands) inside an fp32 torch
forward by rounding chosen tensors ("sites") to fp16, and reports the rel-L2 of the final logits against the un-rounded forward for several
site sets. Weights are drawn per layer with synthetic.llama_state_dict's scales and dropped again (13 B fp32 parameters do not fit next to the
variants); all variants see the same weights (already fp16-representable: a real checkpoint is 16-bit).

    python tools/llm_error_budget.py [--layers 40] [--tokens 64]
"""
import argparse
import math
import time

import torch
import torch.nn.functional as F

SITES = ("n1", "q0", "k0", "q", "k", "v", "p", "ao", "n2", "glu", "lm")    # q0 / k0: the qkv GEMM's 16-bit output ahead of RoPE


def r2(x, mode, dt):
    """mode 0: exact; 1: one 16-bit rounding; 2: hi + lo planes (both 16-bit, lo flushed to zero when subnormal like a denorm-flushing MFMA)."""
    if mode == 0:
        return x
    hi = x.to(dt)
    if mode == 1:
        return hi.float()
    lo = (x - hi.float()).to(dt)
    if dt == torch.float16:
        lo = torch.where(lo.abs() < 6.104e-5, torch.zeros_like(lo), lo)
    return hi.float() + lo.float()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=40)
    ap.add_argument("--tokens", type=int, default=64)
    ap.add_argument("--hidden", type=int, default=5120)
    ap.add_argument("--inter", type=int, default=13824)
    ap.add_argument("--heads", type=int, default=40)
    ap.add_argument("--vocab", type=int, default=8192)
    ap.add_argument("--dtype", default="fp16")
    a = ap.parse_args()
    dt = torch.float16 if a.dtype == "fp16" else torch.bfloat16
    H, I, nh, T, L = a.hidden, a.inter, a.heads, a.tokens, a.layers
    hd = H // nh
    torch.manual_seed(0)
    all1 = {s: 1 for s in SITES}
    none = {s: 0 for s in SITES}
    V = {
        "exact": dict(none),
        "all sites 1x16-bit (the shipped path)": dict(all1),
        "GEMM operands hi+lo (n1 n2 ao glu lm), attention 16-bit": dict(all1, n1=2, n2=2, ao=2, glu=2, lm=2),
        " + v exact-ish (hi+lo)": dict(all1, n1=2, n2=2, ao=2, glu=2, lm=2, v=2),
        " + v, p hi+lo": dict(all1, n1=2, n2=2, ao=2, glu=2, lm=2, v=2, p=2),
        " + q, k hi+lo, v p 16-bit": dict(all1, n1=2, n2=2, ao=2, glu=2, lm=2, q=2, k=2, q0=2, k0=2),
        " + q, k, v hi+lo (fp32-grade KV cache), p 16-bit": dict(all1, n1=2, n2=2, ao=2, glu=2, lm=2, q=2, k=2, q0=2, k0=2, v=2),
        "everything hi+lo": {s: 2 for s in SITES},
        "only n1 16-bit": dict(none, n1=1), "only q0 k0": dict(none, q0=1, k0=1), "only q": dict(none, q=1), "only k": dict(none, k=1), "only v": dict(none, v=1),
        "only p": dict(none, p=1), "only ao": dict(none, ao=1), "only n2": dict(none, n2=1), "only glu": dict(none, glu=1),
        "only lm": dict(none, lm=1),
    }
    x0 = torch.randn(T, H) * 0.5
    X = {k: x0.clone() for k in V}
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))
    fr = torch.outer(torch.arange(T).float(), inv)
    emb = torch.cat((fr, fr), -1)
    cos, sin = emb.cos().to(dt).float()[None], emb.sin().to(dt).float()[None]
    rot = lambda t: torch.cat((-t[..., hd // 2:], t[..., :hd // 2]), -1)
    ii = torch.arange(T)
    causal = ii[None, :] > ii[:, None]
    lin = lambda o, i, g=1.0: (torch.randn(o, i) * (g / math.sqrt(i))).to(dt).float()
    gam = lambda n: 1.0 + 0.1 * torch.randn(n)
    rms = lambda x, w: w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5))
    t0 = time.time()
    for li in range(L):
        wq, wk, wv, wo = lin(H, H, 1.5), lin(H, H, 1.5), lin(H, H, 1.5), lin(H, H, 0.5)
        wg, wu, wd = lin(I, H), lin(I, H), lin(H, I, 0.5)
        g1, g2 = gam(H), gam(H)
        for name, m in V.items():
            x = X[name]
            h = r2(rms(x, g1), m["n1"], dt)
            q = r2(F.linear(h, wq).view(T, nh, hd).transpose(0, 1), m["q0"], dt)
            k = r2(F.linear(h, wk).view(T, nh, hd).transpose(0, 1), m["k0"], dt)
            v = r2(F.linear(h, wv).view(T, nh, hd).transpose(0, 1), m["v"], dt)
            q = r2(q * cos + rot(q) * sin, m["q"], dt)
            k = r2(k * cos + rot(k) * sin, m["k"], dt)
            s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
            s = s.masked_fill(causal, float("-inf"))
            e = torch.exp(s - s.amax(-1, keepdim=True))
            den = e.sum(-1, keepdim=True)                   # flash kernels sum the fp32 exponentials, the PV operand is rounded
            o = (r2(e, m["p"], dt) @ v) / den
            o = r2(o.transpose(0, 1).reshape(T, H), m["ao"], dt)
            x = x + F.linear(o, wo)
            h = r2(rms(x, g2), m["n2"], dt)
            g = r2(F.silu(F.linear(h, wg)) * F.linear(h, wu), m["glu"], dt)
            X[name] = x + F.linear(g, wd)
        if li % 5 == 4 or li == L - 1 or li == 1:
            ref = X["exact"]
            print(f"after layer {li + 1:2d} ({time.time() - t0:5.0f} s): residual-stream rel-L2 " +
                  "  ".join(f"{((X[k] - ref).norm() / ref.norm()).item():.2e}" for k in list(V)[1:8]), flush=True)
    gn, wl = gam(H), lin(a.vocab, H, 2.0)
    ref = F.linear(rms(X["exact"], gn), wl)
    print(f"\nlogits rel-L2 vs the un-rounded forward, {L} layers, {T} tokens, {a.dtype}:")
    for name, m in V.items():
        y = F.linear(r2(rms(X[name], gn), m["lm"], dt), wl)
        print(f"  {name:70s} {((y - ref).norm() / ref.norm()).item():.3e}")


if __name__ == "__main__":
    main()
