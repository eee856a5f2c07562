"""Same-box A/B of the LLM phases of the headline step, precise mode (fp32-grade activations, csrc/precise.hip) vs the plain 16-bit flow:
165-token prefill of 16 requests as one pass, the graph-replayed token step at ~230 keys, the forced 65-token image chunk.
    python tools/bench_llm_precise_ab.py [--dtype fp16]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="fp16")
a = ap.parse_args()
dt = torch.float16 if a.dtype == "fp16" else torch.bfloat16
bench.BATCH = 16
dev = torch.device("cuda:0")


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with torch.no_grad():
    for precise in ("1", "0", "1", "0"):
        os.environ["SX_LLM_PRECISE"] = precise
        _, agent, _ = bench.build_models(dev, dt, need=("llm",))
        llm = agent.llm
        assert llm.precise == (precise == "1")
        G, H = llm.G, llm.config.hidden_size
        xs = [torch.randn(165, H, device=dev) * 0.5 for _ in range(G)]
        ch = [torch.randn(65, H, device=dev) * 0.5 for _ in range(G)]

        def prefill():
            llm.reset()
            llm.forward_embeds_batch(xs, list(range(G)))

        def chunk():
            llm._P["pos"].fill_(230)
            llm._P["ctx"].fill_(231)
            llm.forward_embeds_batch(ch, list(range(G)), need_logits=False)
        t_pre, t_chunk = timed(prefill, 3), timed(chunk, 3)
        ids = torch.full((G, 200), -1, dtype=torch.int32, device=dev)
        hid = torch.zeros((G, 200, H), device=dev)
        img_ids = torch.arange(llm.V - 200, llm.V - 134, dtype=torch.int32, device=dev)

        def setpos():
            llm._P["pos"].fill_(230); llm._P["ctx"].fill_(231); llm._P["step"].zero_(); llm._P["cur"].fill_(5)
        setpos()
        for _ in range(3):
            llm.decode_step(img_ids, ids, hid, use_graph=True)
        setpos()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(62):
            llm.decode_step(img_ids, ids, hid, use_graph=True)
        e1.record()
        torch.cuda.synchronize()
        t_tok = e0.elapsed_time(e1) / 62
        print(f"precise={precise} {a.dtype}: prefill 16 x 165 {t_pre:.1f} ms | token step (16 seqs, 230-292 keys, graph) {t_tok:.3f} ms | "
              f"65-token chunk x 16 {t_chunk:.1f} ms | LLM per headline step ~ {t_pre + 62 * t_tok + t_chunk:.0f} ms", flush=True)
        del agent, llm
        torch.cuda.empty_cache()
