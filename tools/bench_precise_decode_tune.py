"""Token step of the precise decode path (16 sequences, graph replay) under sx_gemv_tune(1, v): 0 = 2 k-steps per round in every two-block
skinny kernel (shipped), 2 = 4 k-steps in the 20-row-tile kernels only (one wave per SIMD there anyway), 1 = 4 k-steps everywhere."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from seedx_amd import _lib

bench.BATCH = 16
dev = torch.device("cuda:0")
lib = _lib.load()
with torch.no_grad():
    _, agent, _ = bench.build_models(dev, torch.float16, need=("llm",))
    llm = agent.llm
    G, H = llm.G, llm.config.hidden_size
    ids = torch.full((G, 200), -1, dtype=torch.int32, device=dev)
    hid = torch.zeros((G, 200, H), device=dev)
    img_ids = torch.arange(llm.V - 200, llm.V - 134, dtype=torch.int32, device=dev)
    for v in (0, 2, 1, 0, 2, 1):
        lib.sx_gemv_tune(1, v)
        llm._graph = None
        llm._P["pos"].fill_(230); llm._P["ctx"].fill_(231); llm._P["step"].zero_(); llm._P["cur"].fill_(5)
        for _ in range(3):
            llm.decode_step(img_ids, ids, hid, use_graph=True)
        llm._P["pos"].fill_(230); llm._P["ctx"].fill_(231); llm._P["step"].zero_()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(62):
            llm.decode_step(img_ids, ids, hid, use_graph=True)
        e1.record()
        torch.cuda.synchronize()
        print(f"sx_gemv_tune(1, {v}): {e0.elapsed_time(e1) / 62:.3f} ms per token step (precise={llm.precise}, 16 sequences)", flush=True)
    lib.sx_gemv_tune(1, 0)
