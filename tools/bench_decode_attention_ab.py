"""Same-process A/B of the decode token step (16 lock-step sequences, HIP-graph replay): fused decode attention vs the
three-kernel form (rope_kv_append + attn_decode + combine).   python tools/bench_decode_attention_ab.py"""
import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
bench.BATCH = 16
dev = torch.device("cuda:0")
with torch.no_grad():
    _, agent, _ = bench.build_models(dev, torch.bfloat16, need=("llm",))
    llm = agent.llm
    G = llm.G
    ids = torch.full((G, 200), -1, dtype=torch.int32, device=dev)
    hid = torch.zeros((G, 200, llm.config.hidden_size), device=dev)
    img_ids = torch.arange(llm.V - 200, llm.V - 134, dtype=torch.int32, device=dev)
    for fused, nsplit in ((False, 8), (False, 2), (True, 8), (True, 1), (False, 1), (False, 2), (True, 1)):
        llm.fused_decode_attention = fused
        llm.decode_nsplit = nsplit
        llm._graph = None
        llm.reset()
        llm._P["pos"].fill_(230); llm._P["ctx"].fill_(231); llm._P["step"].zero_(); llm._P["cur"].fill_(5)
        for _ in range(3):
            llm.decode_step(img_ids, ids, hid, use_graph=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(64):
            llm.decode_step(img_ids, ids, hid, use_graph=True)
        e1.record(); torch.cuda.synchronize()
        print("fused" if fused else "three kernels", "nsplit", nsplit, "%.3f ms per token (16 sequences, ~260 keys)" % (e0.elapsed_time(e1) / 64), flush=True)
